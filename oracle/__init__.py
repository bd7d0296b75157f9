"""CPU oracle for the MERLOT Reserve pretraining step.

TEST INFRASTRUCTURE ONLY.  Nothing in ``merlot_reserve_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the reported CPU baseline.

PARITY UNPINNED: the reference (rowanz/merlot_reserve) ships no tests, no golden
vectors and no recorded outputs, and its implementation (JAX/Flax) cannot be
imported in the build container (jax, flax, optax, clu, tensorflow are absent).
The oracle is therefore a restatement of the reference's algorithm, pinned only
by hand-derived known-answer vectors (tests/test_oracle_known_answers.py), by the
tokenizer fixture generated from the reference's own ``lowercase_encoder`` and by
agreement between two independent restatements (numpy fp64 in ``ref_numpy`` and
torch in ``ref_torch``).
"""
