"""CPU oracle for the DDIM/EDM + NLC sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU (f32 / f64) restatement of the
reference's algorithm (Walleclipse/Diffusion-NLC, every function cites the reference file:line
it follows).  It is imported only by tests/, by __graft_entry__.smoke() and by the cpu_baseline
leg of bench.py - as the checker, never as the thing measured or shipped.  The product path
(diffusion-nlc_amd/) never imports it and has no CPU fallback.

Pinning: tests/golden/*.npz hold outputs of the reference itself (imported from
/root/reference in the build container by tests/golden/make_golden.py); tests/test_oracle_golden.py
checks every oracle function against them.
"""
