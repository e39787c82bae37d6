#!/usr/bin/env python
"""Soak: optimize_embedding (public API) for N steps at SD-1.5 shapes; prints loss trajectory and memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablekeypoints_amd.optimize import optimize_embedding, default_args
from stablekeypoints_amd.optimize_token import load_ldm
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ldm, controllers, n = load_ldm("cuda:0", "sd15", feature_upsample_res=128)
args = default_args(num_tokens=77, feature_upsample_res=128, batch_size=4, num_steps=steps, device="cuda:0", log_interval=5, max_len=32)
torch.manual_seed(0)
ctx0 = torch.randn(1, 77, 768)
t0 = time.time()
out = optimize_embedding(ldm, args, controllers, n, context=ctx0.clone())
torch.cuda.synchronize()
print("seconds", time.time() - t0, "finite", bool(torch.isfinite(out).all()), "max|delta|", float((out.cpu() - ctx0).abs().max()),
      "peak GB", torch.cuda.max_memory_allocated() / 2**30, "reserved GB", torch.cuda.memory_reserved() / 2**30)
