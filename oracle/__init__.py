"""oracle/ -- CPU restatement (numpy, fp32) of the reference's rule-guided sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under this directory is shipped or measured as the
product: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it,
and only as the *checker*.  The product path (rule-guided-music_amd/) never imports oracle/
and fails loudly when the HIP library is missing.

How it is pinned:
  * the reference ships no tests / golden vectors of its own, so every function here is pinned
    against outputs of the reference itself, produced IN THE BUILD CONTAINER by importing
    /root/reference (tests/golden/make_golden.py, committed) and stored as small .npz fixtures
    under tests/golden/;  tests/test_oracle_golden.py replays them on CPU.
  * PARITY UNPINNED slice: timm==0.9.2 `Mlp` and rotary-embedding-torch==0.3.2
    `RotaryEmbedding.rotate_queries_or_keys` are un-vendored third-party packages, absent from
    /root/reference and from this image.  Their published algorithm is restated here
    (dit_np.rotary_tables / dit_np.mlp) and in tests/golden/ref_shims.py; the goldens therefore
    pin everything AROUND them (the reference's own call sites dit.py:263-288, :326) but not
    the two packages' internals.  Likewise music21/mido (chord rule) cannot be restated.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
