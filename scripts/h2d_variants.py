"""Which piece of the host-fed step costs time: the graph-replayed base step timed with the inputs resident, fed by loader.PrefetchLoader, and
fed by stripped variants of the loader (no H2D copy / no host memcpy / copy on the compute stream / one host thread / no event waits)."""
import os, sys, time, itertools, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_reserve_amd import config as cfg, synthetic
from merlot_reserve_amd.trainer import Trainer
from merlot_reserve_amd import loader as L

dev = torch.device('cuda:0')
c = cfg.load_config('base')
B = 4
tr = Trainer(c, B, dev)
batches = [synthetic.make_batch(c, B, seed=1234 + i, device=dev) for i in range(2)]
plans = [tr.plan(b) for b in batches]
tr.train_step(batches[0], plan=plans[0])
tr.capture(batches[0])
host = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
N = int(os.environ.get('N', '24'))

def sync(): torch.cuda.synchronize()

def resident():
    for i in range(3): tr.train_step_graph(batches[i % 2], plans[i % 2])
    sync(); t = time.perf_counter()
    for i in range(N): tr.train_step_graph(batches[i % 2], plans[i % 2])
    sync(); return (time.perf_counter() - t) / N * 1e3


class Variant(L.PrefetchLoader):
    no_h2d = no_memcpy = main_stream = no_wait = False

    def _issue(self):
        try:
            batch = next(self.it)
        except StopIteration:
            return False
        slot = self.turn
        self.turn = (self.turn + 1) % self.depth
        if self.pinned[slot] is None:
            self._alloc(slot, batch)
        if self.h2d_done[slot] is not None:
            self.h2d_done[slot].synchronize()
        if not self.no_memcpy:
            for k in L.FLOAT_KEYS:
                self.pinned[slot][k].copy_(batch[k])
        st = torch.cuda.current_stream() if self.main_stream else self.copy_stream
        with torch.cuda.stream(st):
            if self.consumed[slot] is not None and not self.no_wait:
                st.wait_event(self.consumed[slot])
            if not self.no_h2d:
                for k in L.FLOAT_KEYS:
                    self.dev[slot][k].copy_(self.pinned[slot][k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        self.h2d_done[slot] = ev
        self.queue.append((slot, {k: v for k, v in batch.items() if k not in L.FLOAT_KEYS}))
        return True

    def __next__(self):
        self.release()
        self._issue()
        if not self.queue:
            raise StopIteration
        slot, rest = self.queue.pop(0)
        if not self.no_wait:
            torch.cuda.current_stream().wait_event(self.h2d_done[slot])
        self.current = slot
        out = dict(rest)
        out.update(self.dev[slot])
        return out


def fed(**flags):
    cls = type('V', (Variant,), flags)
    ld = cls(itertools.islice(itertools.cycle(host), N + 3), dev, depth=int(os.environ.get('DEPTH', '2')))
    ht = 0.0
    for i, b in enumerate(ld):
        if i == 3:
            sync(); t = time.perf_counter(); ht = 0.0
        t0 = time.perf_counter()
        tr.train_step_graph(b, plans[i % 2])
        ht += time.perf_counter() - t0
    sync()
    return (time.perf_counter() - t) / N * 1e3, ht / N * 1e3


def stock():
    ld = L.PrefetchLoader(itertools.islice(itertools.cycle(host), N + 3), dev, depth=2)
    for i, b in enumerate(ld):
        if i == 3:
            sync(); t = time.perf_counter()
        tr.train_step_graph(b, plans[i % 2])
    sync()
    return (time.perf_counter() - t) / N * 1e3

for rnd in range(2):
    print(f'--- round {rnd}', flush=True)
    print(f'resident                      {resident():7.2f} ms', flush=True)
    print(f'stock PrefetchLoader          {stock():7.2f} ms', flush=True)
    for name, fl in (('variant = stock', {}), ('no H2D copy', dict(no_h2d=True)), ('no host memcpy', dict(no_memcpy=True)),
                     ('no H2D, no memcpy', dict(no_h2d=True, no_memcpy=True)), ('copy on the compute stream', dict(main_stream=True)),
                     ('no event waits', dict(no_wait=True)), ('no waits, no memcpy', dict(no_wait=True, no_memcpy=True))):
        ms, ht = fed(**fl)
        print(f'{name:29s} {ms:7.2f} ms   (host time in train_step_graph {ht:5.2f})', flush=True)
    torch.set_num_threads(1)
    ms, ht = fed()
    print(f'stock, 1 host thread          {ms:7.2f} ms   (host time in train_step_graph {ht:5.2f})', flush=True)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    print(f'resident again                {resident():7.2f} ms', flush=True)
