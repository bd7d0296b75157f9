"""Where do a kernel's register spills sit?  For every kernel of build/<file>.s with spilled VGPRs: the basic blocks that contain scratch traffic, with their
MFMA counts and whether a backward branch encloses them (= inside a loop).   python scripts/spill_sites.py [file.s ...]"""
import os, re, sys
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'merlot_reserve_amd', 'build')
files = sys.argv[1:] or [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith('.s')]
for path in files:
    text = open(path).read().splitlines()
    # kernels: from "<name>:" at column 0 following .type <name>,@function up to .Lfunc_end
    i = 0
    while i < len(text):
        m = re.match(r'^(_Z\w+):\s*(;.*)?$', text[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < len(text) and not text[j].startswith('.Lfunc_end'):
            j += 1
        body = text[i + 1:j]
        i = j
        if not any('scratch_' in l for l in body):
            continue
        # basic blocks
        blocks, cur = [], ['<entry>', 0, 0, 0, 0]
        label_at = {}
        for n, l in enumerate(body):
            lm = re.match(r'^(\.LBB\d+_\d+):', l)
            if lm:
                blocks.append(cur)
                cur = [lm.group(1), 0, 0, 0, n]
                label_at[lm.group(1)] = n
                continue
            s = l.strip()
            if s.startswith('v_mfma'):
                cur[1] += 1
            elif s.startswith('scratch_store'):
                cur[2] += 1
            elif s.startswith('scratch_load'):
                cur[3] += 1
        blocks.append(cur)
        loops = []      # (start line, end line) of backward branches
        for n, l in enumerate(body):
            bm = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s*s_branch\s+(\.LBB\d+_\d+)', l)
            if bm:
                tgt = bm.group(1) or bm.group(2)
                if tgt in label_at and label_at[tgt] < n:
                    loops.append((label_at[tgt], n))
        print(f'{os.path.basename(path)}: {name[:90]}')
        for b in blocks:
            if b[2] or b[3]:
                enc = sorted([(e - s, s, e) for s, e in loops if s <= b[4] <= e])
                where = 'not in a loop'
                if enc:
                    def nmf(s0, e0):
                        return sum(1 for l in body[s0:e0 + 1] if l.strip().startswith('v_mfma'))
                    # a k-loop = a loop with MFMAs that contains no smaller loop with MFMAs
                    kl = [(s0, e0) for _, s0, e0 in enc if nmf(s0, e0) and not any(s0 <= s1 and e1 <= e0 and (s1, e1) != (s0, e0) and nmf(s1, e1) for s1, e1 in loops)]
                    _, s0, e0 = enc[0]
                    where = (f'IN THE K-LOOP ({kl[0][1] - kl[0][0]} lines, {nmf(*kl[0])} MFMAs)' if kl else
                             f'in an outer loop only (innermost: {e0 - s0} lines, {nmf(s0, e0)} MFMAs; depth {len(enc)})')
                print(f'    block {b[0]:12s} mfma {b[1]:4d}  scratch stores {b[2]:3d} loads {b[3]:3d}  {where}')
