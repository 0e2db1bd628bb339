"""oracle/ — CPU restatement of the reference algorithms on the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under xva-trainer_amd/ may import this package; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
checker / reported baseline — never as the thing shipped or measured as the product.

Every function cites the reference file:line it restates (paths relative to /root/reference).
The restatement is pinned against the reference itself: oracle/gen_golden.py imports the
reference's Python modules in the build container (with stubbed librosa/numba, see
oracle/ref_import.py), runs them on seeded inputs and writes tests/golden/*.npz;
tests/test_oracle_golden.py checks this package against those vectors.

Parity-unpinned pieces (third-party arithmetic that is NOT in /root/reference and has no
reference test or vector): librosa==0.8.1 filters.mel (Slaney filterbank) and
librosa.util.normalize — restated here from the published algorithm; the reference run used
to make the golden vectors necessarily uses the same restatement as its librosa stub.
"""
