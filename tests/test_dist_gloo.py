"""world_size-2 gloo test of the data-parallel exchange step: rank-sharded images,
SUM all-reduce of the embedding gradient, identical Adam state on every rank."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from stablekeypoints_amd import dist as D
    w, r, _ = D.init_from_env("gloo")
    assert (w, r) == (world, rank) and D.world_size() == world
    torch.manual_seed(100 + rank)                                   # ranks start from DIFFERENT embeddings
    ctx = torch.randn(1, 7, 768, requires_grad=True)
    opt = torch.optim.Adam([ctx], lr=5e-3)
    red = D.EmbeddingReducer(ctx, opt)                              # broadcast from rank 0
    gen = torch.Generator().manual_seed(7)
    per_image = torch.randn(3, 4, 1, 7, 768, generator=gen)         # [step, image, ...] global batch 4
    hist = []
    for step in range(3):
        mine = D.shard_indices(list(range(4)), rank, world)
        for i in mine:                                              # local accumulation
            (ctx * per_image[step, i]).sum().div(4).backward()
        red.step()
        hist.append(ctx.detach().clone())
    torch.save(torch.stack(hist), os.path.join(out, f"r{rank}.pt"))
    D.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b)                                        # bit-identical embeddings on every rank
    torch.manual_seed(100)
    ctx = torch.randn(1, 7, 768, requires_grad=True)
    opt = torch.optim.Adam([ctx], lr=5e-3)
    per_image = torch.randn(3, 4, 1, 7, 768, generator=torch.Generator().manual_seed(7))
    for step in range(3):
        for i in range(4):
            (ctx * per_image[step, i]).sum().div(4).backward()
        opt.step(); opt.zero_grad()
        torch.testing.assert_close(a[step], ctx.detach(), rtol=1e-6, atol=1e-7)


def _aug_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from stablekeypoints_amd import dist as D
    from stablekeypoints_amd.eval import finish_augmented
    D.init_from_env("gloo")
    g = torch.Generator().manual_seed(3)
    cover = (torch.rand(4, 2, 5, 5, generator=g) > 0.6).float()                    # pixels a view covers
    vals = torch.rand(4, 2, 5, 5, generator=g) * cover                             # uncovered pixels contribute 0
    tot, num = vals[rank * 2:(rank + 1) * 2].sum(0), cover[rank * 2:(rank + 1) * 2].sum(0)   # this rank's two views
    torch.save(finish_augmented(tot, num), os.path.join(out, f"a{rank}.pt"))
    D.barrier()
    dist.destroy_process_group()


def test_sharded_augmented_inference_reduction(tmp_path):
    """augmentations sharded over ranks, one all-reduce of (sum, count): same maps on every rank and
    equal to the single-process result over all views; 0/0 pixels become 0."""
    port = 29300 + (os.getpid() % 150)
    mp.spawn(_aug_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "a0.pt"), torch.load(tmp_path / "a1.pt")
    assert torch.equal(a, b)
    g = torch.Generator().manual_seed(3)
    cover = (torch.rand(4, 2, 5, 5, generator=g) > 0.6).float()
    vals = torch.rand(4, 2, 5, 5, generator=g) * cover
    tot, num = vals.sum(0), cover.sum(0)
    assert (num == 0).any()                                     # some pixels are covered by no view at all: 0/0 -> 0
    ref = tot / num
    ref[ref != ref] = 0
    torch.testing.assert_close(a, ref, rtol=1e-6, atol=1e-7)
    assert torch.isfinite(a).all()


def test_shard_indices_partition():
    from stablekeypoints_amd.dist import shard_indices
    perm = list(range(11))
    parts = [shard_indices(perm, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(8)) and all(len(p) == 2 for p in parts)
