"""GPU tests added in round 5.

* `python bench.py --gpus 2` WITHOUT a launcher around it (the N > 1 line must not depend on who wraps the command);
* RCCL itself on the box: `init_process_group("nccl", world_size=1)`, the embedding-gradient all-reduce ordered against the
  step's HIP kernels, 1000 calls timed (tools/rccl_smoke.py);
* the protocol bench of BASELINE config 2 (`optimize_embedding` as the reference runs it) on a short, reduced-width run:
  host-resident folder data through the one-batch-ahead loader gives the SAME embedding as the synchronous loop.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_two_ranks_one_gpu():
    """No torchrun, no WORLD_SIZE: bench.py re-runs itself under torch.distributed.run (both ranks on cuda:0, gloo standing in
    for RCCL on the one-GPU box), rank 0 prints the one JSON line, exit status 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SKP_DIST_BACKEND="gloo", SKP_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--image-size", "128", "--res", "32",
           "--tokens", "16", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off", "--kernel-iters", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = _json_lines(out)
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert d["collective_check"]["ok"] and d["collective_check"]["ranks_seen"] == 2
    assert d["collective_check"]["embedding_identical_on_all_ranks"] is True
    assert "torch.distributed.run" in out.stderr


def test_rccl_single_rank_allreduce_on_step_stream():
    """RCCL on the box (a13 / (e)): communicator init, the [1,77,768] gradient all-reduced 1000x on the step's stream, the
    value, the ordering against the step's kernels, us per call."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py"), "--calls", "1000"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    lines = _json_lines(out)
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    d = lines[0]
    print("RCCL single-rank:", json.dumps(d))
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["elements"] == 77 * 768
    assert d["value_ok"] and d["ordered_against_step_kernels"]
    assert 0 < d["us_per_call_stream"] < 500


def test_optimize_embedding_host_folder_prefetch_is_order_identical(tmp_path):
    """`optimize_embedding` on a HOST-resident folder of PNG files (`dataset_name="custom"`, reference
    datasets/custom_images.py) through the one-group-ahead loader (decode threads + pinned staging + async H2D) against
    the synchronous loop: the same images in the same order, so the embedding after every optimizer step is bit-identical;
    the step callback fires once per optimizer step."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from protocol_bench import write_png_folder
    from stablekeypoints_amd.optimize import default_args, optimize_embedding
    from stablekeypoints_amd.optimize_token import load_ldm
    folder = write_png_folder(str(tmp_path / "imgs"), 6, 128, seed=4)
    ldm, controllers, n = load_ldm("cuda", "tiny", feature_upsample_res=32)
    trajs = {}
    for workers in (0, 3):
        args = default_args(dataset_name="custom", dataset_loc=folder, num_steps=4, batch_size=2, num_tokens=16,
                            feature_upsample_res=32, furthest_point_num_samples=8, top_k=4, image_size=128, device="cuda",
                            log_interval=0, loader_workers=workers, seed=5)
        torch.manual_seed(11)                                       # the affine / noise draws of the loop
        torch.cuda.manual_seed(11)
        traj, calls = [], []
        ctx0 = torch.randn(1, 16, 768, generator=torch.Generator().manual_seed(2)) * 3.0
        out = optimize_embedding(ldm, args, controllers, n, context=ctx0.clone(), trajectory_out=traj, step_callback=calls.append)
        assert calls == [0, 1, 2, 3] and torch.equal(out, traj[-1])
        trajs[workers] = torch.cat(traj).cpu()
    assert not torch.equal(trajs[0][0], trajs[0][-1])
    assert torch.equal(trajs[0], trajs[3]), "the prefetching loader changed the images or their order"
