"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (PyTorch fp32 / fp64, eager) restatement of the StableKeypoints token-optimisation
hot path, written from the behaviour of the reference (ubc-vision/StableKeypoints) and
pinned against golden vectors that `oracle/gen_golden.py` captured by importing the
reference's own functions in the build container (fixtures in `tests/golden/`).

Rules (enforced by tests/test_layout.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
    anything from this package -- and only as the checker / the timed CPU baseline;
  * nothing under `stablekeypoints_amd/` imports it; the product path has no CPU fallback
    and raises when the HIP library is missing.

Parity status: pinned at the operator boundary (hook, collect_maps, selection, losses,
affine warp, hook-subgraph gradient) by goldens G1..G7.  The surrounding UNet/VAE are
"parity unpinned" (diffusers==0.8.0 and the SD weights are not available offline; see
DESIGN.md section 3).
"""
