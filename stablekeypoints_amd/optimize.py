"""Token optimisation loop and losses, reference API of optimize.py (collect_maps, sharpening_loss,
equivariance_loss, find_gaussian_loss_at_point, optimize_embedding).

`optimize_embedding` keeps the reference's signature and per-optimizer-step semantics
(optimize.py:269-452) but is laid out for one MI355X per process:
  * the image and its affine copy go through the VAE/UNet as ONE batch, and all images of an
    accumulation group are batched together (the frozen network sees B = 2*images rows; the
    gradient of the shared embedding sums over rows, which is exactly what the reference's
    accumulation loop + `.repeat` backward compute, optimize.py:420-425, ptp_utils.py:229);
  * the forward stops after the 4th hooked layer (result-identical);
  * maps, arg-max, KL ranking, furthest-point sampling and both losses run in HIP kernels with no
    host synchronisation; the only collective is one all-reduce of the [1,T,768] gradient per step.
"""
from __future__ import annotations

import time
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import dist as skp_dist
from . import ops, ptp_utils
from ._maps import collect_maps, collect_maps_batched          # noqa: F401  (collect_maps is public API)
from .eval import find_k_max_pixels
from .invertable_transform import RandomAffineWithInverse
from .optimize_token import gaussian_circles


# ---------------------------------------------------------------------------------------------
# losses, reference-shaped API                                         optimize.py:157-206
# ---------------------------------------------------------------------------------------------
def _all_rows(n, device):
    return torch.arange(n, device=device, dtype=torch.int64)


def sharpening_loss(attn_map, sigma=1.0, temperature=1e1, device="cuda", num_subjects=1):
    """optimize.py:166-179 on the fused loss kernel (identity affine => the equivariance term is unused)."""
    am, _ = ops.token_stats(attn_map, num_subjects=num_subjects, sigma=sigma, want_kl=False)
    sharp, _ = ops.fused_losses(attn_map, attn_map, _all_rows(attn_map.shape[0], attn_map.device), am,
                                [1, 0, 0, 0, 1, 0], sigma, num_subjects)
    return sharp


def find_gaussian_loss_at_point(attn_map, pos, sigma=1.0, temperature=1e-1, device="cuda", indices=None,
                                num_subjects=1):
    """optimize.py:182-206 for caller-supplied positions (torch ops; not on the optimisation path)."""
    target = gaussian_circles(pos, size=attn_map.shape[1], sigma=sigma).to(attn_map.device)
    if indices is not None:
        attn_map, target = attn_map[indices], target[indices]
    return F.mse_loss(attn_map, target)


def equivariance_loss(embeddings_initial, embeddings_transformed, transform, index):
    """optimize.py:157-163: MSE(map, unwarp(map_T)[index]) with transform.last_params['theta'][index]."""
    theta = transform.last_params["theta"][index].reshape(-1).tolist()
    mt = embeddings_transformed[index] if embeddings_transformed.dim() == 4 else embeddings_transformed
    k = embeddings_initial.shape[0]
    am = torch.zeros(1, k, device=embeddings_initial.device, dtype=torch.int32)
    _, equiv = ops.fused_losses(embeddings_initial, mt, _all_rows(k, embeddings_initial.device), am, theta, 1.0, 1)
    return equiv


# ---------------------------------------------------------------------------------------------
# datasets: only the {'img': float[3,H,W] in [0,1]} contract matters on this path
# ---------------------------------------------------------------------------------------------
class SyntheticImages(torch.utils.data.Dataset):
    """`torch.rand(N,3,S,S)` under a fixed seed (SURVEY.md 8(d) synthetic inputs), resident on `device`."""

    def __init__(self, n=64, size=512, seed=0, device="cpu"):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.data = torch.rand(n, 3, size, size, generator=g).to(device)

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        return {"img": self.data[i]}


def build_dataset(args):
    name = getattr(args, "dataset_name", "synthetic")
    if name == "synthetic":
        return SyntheticImages(n=getattr(args, "max_len", -1) if getattr(args, "max_len", -1) > 0 else 64,
                               size=getattr(args, "image_size", 512), seed=getattr(args, "seed", 0),
                               device=getattr(args, "device", "cpu"))
    if name == "custom":
        from .custom_images import CustomDataset
        return CustomDataset(data_root=args.dataset_loc, image_size=getattr(args, "image_size", 512))
    raise NotImplementedError(f"dataset '{name}' is outside the hot-path scope (SURVEY.md 2.1 row 15); "
                              "use 'synthetic' or 'custom'")


# ---------------------------------------------------------------------------------------------
# host input pipeline: the reference's DataLoader iteration, one group ahead          optimize.py:333-347
# ---------------------------------------------------------------------------------------------
def _group_indices(args, n_items, draws, accum, group, rank, world):
    """Generator of (iteration counter, dataset indices) for successive groups of this rank, in the loop's order:
    `draws[0]` when the caller injects the order (parity tests), otherwise epoch-wise shuffled, rank-sharded indices from a
    generator seeded by `args.seed` (every rank draws the same permutation and takes elements rank, rank+world, ...)."""
    shuffle_gen = torch.Generator(device="cpu").manual_seed(getattr(args, "seed", 0) + 1234)
    order, cursor = [], 0
    for step in range(int(args.num_steps)):
        done = 0
        while done < accum:
            n = min(group, accum - done)
            it = step * accum + done
            idx = [int(i) for i in draws[0][it:it + n]] if draws is not None else []
            while len(idx) < n:
                if cursor >= len(order):
                    perm = torch.randperm(n_items, generator=shuffle_gen).tolist()
                    order, cursor = skp_dist.shard_indices(perm, rank, world), 0
                    if not order:
                        raise ValueError("dataset smaller than the data-parallel width")
                idx.append(order[cursor]); cursor += 1
            yield it, idx
            done += n


class GroupLoader:
    """Iterator of (iteration counter, indices, images [n,3,H,W]) over `index_iter`.

    With `workers` > 0 and a host-resident dataset the NEXT group is decoded while the GPU works on the current one:
    `workers` threads call `dataset[i]["img"]` (file decode + resize release the GIL), one assembler thread stacks the
    items into page-locked memory, and the launch thread only queues one asynchronous H2D copy on its stream (12.6 MB per
    4 x 512^2 group).  The pinned block goes back to torch's host allocator, which recycles it only after the copy's event
    has completed, so the launch thread may run any number of steps ahead of the GPU.  A device-resident dataset
    (tensors already on the GPU) or `workers=0` is the synchronous loop: nothing to hide.  The sequence of indices and the
    pixel values are identical in both modes."""

    def __init__(self, dataset, index_iter, device, workers=4):
        self.dataset, self.it, self.device = dataset, iter(index_iter), torch.device(device)
        self.workers = max(0, int(workers))
        self.pool = self.assembler = self.pending = None
        if self.workers and self.device.type == "cuda":
            probe = dataset[0]["img"]
            if torch.is_tensor(probe) and probe.device.type == "cpu":
                from concurrent.futures import ThreadPoolExecutor
                self.pool = ThreadPoolExecutor(self.workers, thread_name_prefix="skp-decode")
                self.assembler = ThreadPoolExecutor(1, thread_name_prefix="skp-assemble")
                self._submit()

    def _load(self, idx):
        items = list(self.pool.map(lambda i: self.dataset[i]["img"], idx))
        return torch.stack(items).pin_memory()

    def _submit(self):
        nxt = next(self.it, None)
        self.pending = None if nxt is None else (nxt, self.assembler.submit(self._load, nxt[1]))

    def __iter__(self):
        return self

    def __next__(self):
        if self.assembler is None:
            it, idx = next(self.it)
            return it, idx, torch.stack([self.dataset[i]["img"] for i in idx])
        if self.pending is None:
            raise StopIteration
        (it, idx), fut = self.pending
        batch = fut.result()
        self._submit()                                           # group g+1 decodes while group g runs on the GPU
        return it, idx, batch.to(self.device, non_blocking=True)

    def close(self):
        if self.assembler is not None:
            if self.pending is not None:
                self.pending[1].cancel()
            self.assembler.shutdown(wait=True)
            self.pool.shutdown(wait=True)
            self.assembler = self.pool = self.pending = None


# ---------------------------------------------------------------------------------------------
# one accumulation group on this rank
# ---------------------------------------------------------------------------------------------
def image_losses(attn_map, attn_map_t, theta_row, args):
    """optimize.py:380-414 for one image, entirely on device: -> (sharp, equiv, selected tokens)."""
    sigma, ns = args.sigma, getattr(args, "num_subjects", 1)
    strategy = getattr(args, "top_k_strategy", "gaussian")
    am, order = token_order(attn_map, strategy, ns, sigma)
    am_t, _ = ops.token_stats(attn_map_t, num_subjects=1, sigma=sigma, want_kl=False)
    n_cand = min(args.furthest_point_num_samples, attn_map.shape[0])
    # the reference's greedy loop stops adding once every candidate is chosen (ptp_utils.py:142-157), i.e. it returns
    # min(top_k, n_cand) tokens
    _, sel = ops.select_tokens(order, am_t[0], attn_map.shape[-1], n_cand, min(args.top_k, n_cand))
    sharp, equiv = ops.fused_losses(attn_map, attn_map_t, sel, am, theta_row, sigma, ns)
    return sharp, equiv, sel


def token_order(attn_map, strategy, num_subjects, sigma):
    """Per-token score whose ascending order gives the candidate tokens (optimize.py:382-393):
    'gaussian' = KL to the gaussian target (ptp_utils.py:86-112), 'entropy' = entropy of the softmax-normalised map
    (ptp_utils.py:165-187), 'consistent' = the token index.  -> (argmax i32 [num_subjects,T], score f32 [T])."""
    if strategy == "gaussian":
        return ops.token_stats(attn_map, num_subjects=num_subjects, sigma=sigma, want_kl=True)
    if strategy == "entropy":
        am, _, ent = ops.token_stats(attn_map, num_subjects=num_subjects, sigma=sigma, want_kl=False, want_entropy=True)
        return am, ent
    if strategy == "consistent":
        am, _ = ops.token_stats(attn_map, num_subjects=num_subjects, sigma=sigma, want_kl=False)
        return am, torch.arange(attn_map.shape[0], device=attn_map.device, dtype=torch.float32)
    raise NotImplementedError(strategy)


token_order._skp_stock_order = True         # ops.MapLossesFn batches the statistics of all images only for this scoring


def _group_losses(controller, thetas, args, n, theta_inv_dev=None):
    """(sum_i equiv_i, sum_i sharp_i) of the 2n stored rows.  Default: ONE autograd node from the hooked layers' q / k to
    the two loss sums (ops.MapLossesFn), whose backward hands the map kernels the gradient as K selected rows per batch
    row -- taken where it is the faster one (T > 128 by default, see ops.MAP_BWD_MODE).  Otherwise the general route:
    maps as a tensor, one loss node per image, dense [2n,T,R,R] map gradient."""
    records = [rec for i, rec in enumerate(controller.step_store["attn"]) if i in args.layers]
    sides = [int(round(r.q.shape[1] ** 0.5)) for r in records]
    T = records[0].k.shape[1]
    n_cand = min(args.furthest_point_num_samples, T)
    K = min(args.top_k, n_cand)
    fused = ops.map_bwd_sparse_supported(sides, K, records[0].R, T, records[0].heads) and K >= 2
    controller.fused_route = bool(fused)                          # GraphedStep captures only steps that take the fused node
    if theta_inv_dev is not None and not fused:
        raise RuntimeError("_group_losses: device-resident affines are served by the fused map + losses node only")
    if fused:
        meta = dict(R=records[0].R, heads=records[0].heads, scales=[r.scale for r in records],
                    thetas=None if theta_inv_dev is not None else [thetas[i].reshape(-1).tolist() for i in range(n)],
                    theta_inv_dev=theta_inv_dev, sigma=args.sigma,
                    num_subjects=getattr(args, "num_subjects", 1), strategy=getattr(args, "top_k_strategy", "gaussian"),
                    n_cand=n_cand, top_k=K, score_fn=token_order)
        flat = []
        for r in records:
            flat += [r.q, r.k]
        controller.reset()
        tot_s, tot_e, _ = ops.MapLossesFn.apply(meta, *flat)
        return tot_e, tot_s
    maps = collect_maps_batched(controller, layers=args.layers)  # [2n,T,R,R]
    dev = maps.device
    tot_e = torch.zeros((), device=dev)
    tot_s = torch.zeros((), device=dev)
    rows = maps.unbind(0)                                        # one backward node: the 2n map gradients are stacked once
    for i in range(n):
        sharp, equiv, _ = image_losses(rows[i], rows[n + i], thetas[i].reshape(-1).tolist(), args)
        tot_e = tot_e + equiv
        tot_s = tot_s + sharp
    return tot_e, tot_s


def _latents_with_cache(ldm, images, warped, cache, ids, dev):
    """Latents of [images; warped] with the un-warped views' rows taken from / added to `cache` (dataset index ->
    [4,h,w]): `image2latent` is the posterior MEAN (ptp_utils.py:289-304), a pure function of the dataset image, so
    from the second epoch on only the warped views go through the VAE encoder."""
    miss = [j for j, i in enumerate(ids) if i not in cache]
    enc = ptp_utils.image2latent(ldm, torch.cat([images[miss], warped], dim=0) if miss else warped, dev)
    for r, j in enumerate(miss):
        cache[ids[j]] = enc[r].clone()
    return torch.cat([torch.stack([cache[i] for i in ids]), enc[len(miss):]], dim=0)


def group_step(ldm, images, context, args, controller, transform, denom, noise=None, thetas=None, latent_cache=None,
               ids=None):
    """Forward both views of `images` [n,3,H,W] as one batch, losses per image, backward of
    sum_i (w_e*equiv_i + w_s*sharp_i)/denom into `context.grad`.  Returns detached (total, equiv, sharp).
    `latent_cache` (dict) + `ids` (the dataset indices of `images`): opt-in reuse of the un-warped views' latents."""
    n = images.shape[0]
    dev = context.device
    images = images.to(dev)
    warped = transform(images, theta=thetas)                     # draws n thetas (4 uniforms each) unless given
    thetas = transform.last_theta_host                           # host copy: no device read-back
    if thetas is None:
        thetas = transform.last_params["theta"].detach().cpu()   # caller passed device thetas
    if latent_cache is not None and ids is not None:
        ptp_utils.find_pred_noise(ldm, None, context, noise_level=args.noise_level, device=dev, noise=noise, early_exit=True,
                                  controllers={dev: controller},
                                  latents=_latents_with_cache(ldm, images, warped, latent_cache, list(ids), dev))
    else:
        both = torch.cat([images, warped], dim=0)
        ptp_utils.find_pred_noise(ldm, both, context, noise_level=args.noise_level, device=dev, noise=noise,
                                  early_exit=True, controllers={dev: controller})
    tot_e, tot_s = _group_losses(controller, thetas, args, n)
    loss = (tot_e * args.equivariance_attn_loss_weight + tot_s * args.sharpening_loss_weight) / denom
    loss.backward()
    return loss.detach(), tot_e.detach() / denom, tot_s.detach() / denom


class GraphedStep:
    """`group_step` as a captured hipGraph (torch.cuda.graphs): forward of both views, losses and the backward into
    `context.grad`, replayed once per group; the all-reduce and Adam stay outside (`EmbeddingReducer.step`).

    Why: a step is ~670-800 launches whatever its batch.  At 4 images per group the launch thread is done in a quarter of the
    step's 76 ms; at 1 image (BASELINE config 3's per-rank shape, B = 2 rows) the same launches have to go out in ~26 ms: the
    thread needs 18-26 ms of host time for them depending on the host (profiles/r06_summary.md: 26.6 ms per step on one box,
    28.2 on a slower one, with eight ranks sharing a host still to come).  Replaying the graph costs 3.4 ms of host time; the
    GPU time is unchanged (26.7 vs 26.4 ms on the fast-host box).

    What it takes: nothing inside the step may come from the host.  The images, the noise of both views, the forward affines
    (for the warp) and their inverses (for the equivariance term: `skp_losses_fwd_dev_f32` reads them from device memory) live
    in static buffers refreshed before every replay -- the thetas are still drawn on the host in the reference's order
    (invertable_transform.py:22-57), the noise by the same device generator call as the eager step, so a seeded run sees the
    same draws in both modes.  Token selection never left the device anyway.  The first `warmup` calls of a group size run
    eagerly (real steps: caches of the frozen weights fill, the route is known); a group size whose step does not take the
    fused map + losses node, or whose capture fails, stays eager (one warning).  One graph (and one private memory pool) per
    group size."""

    def __init__(self, ldm, context, args, controller, transform, denom, warmup=2, warn_route=True):
        self.ldm, self.context, self.args, self.controller, self.transform = ldm, context, args, controller, transform
        self.denom, self.warmup, self.warn_route = denom, int(warmup), bool(warn_route)
        self.dev = context.device
        self.state = {}                                           # group size -> dict(calls, graph, buffers) or "eager"

    def _body(self, st):
        n, dev, args = st["n"], self.dev, self.args
        images = st["images"]
        warped = F.grid_sample(images, F.affine_grid(st["theta"], images.size(), align_corners=False), align_corners=False)
        ptp_utils.find_pred_noise(self.ldm, torch.cat([images, warped], dim=0), self.context, noise_level=args.noise_level,
                                  device=dev, noise=st["noise"], early_exit=True, controllers={dev: self.controller})
        tot_e, tot_s = _group_losses(self.controller, None, args, n, theta_inv_dev=st["theta_inv"])
        loss = (tot_e * args.equivariance_attn_loss_weight + tot_s * args.sharpening_loss_weight) / self.denom
        loss.backward()
        st["out"].copy_(torch.stack([loss.detach(), tot_e.detach() / self.denom, tot_s.detach() / self.denom]))

    def _capture(self, st, images):
        n, dev = st["n"], self.dev
        with torch.no_grad():
            lat = ptp_utils.image2latent(self.ldm, images[:1].to(dev), dev)       # latent geometry of this image size
        st["images"] = torch.empty(n, *images.shape[1:], device=dev, dtype=torch.float32)
        st["theta"] = torch.empty(n, 2, 3, device=dev)
        st["theta_inv"] = torch.empty(n, 6, device=dev)
        st["noise"] = torch.empty(2 * n, *lat.shape[1:], device=dev)
        st["out"] = torch.zeros(3, device=dev)
        # valid contents while the graph is recorded (zeros for the noise: recording must not consume the generator)
        self._stage(st, images, torch.eye(2, 3).repeat(n, 1, 1), torch.zeros_like(st["noise"]))
        if self.context.grad is None:
            self.context.grad = torch.zeros_like(self.context)    # the captured backward ACCUMULATES into this tensor
        st["grad"] = self.context.grad
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._body(st)
        st["graph"] = graph

    def _stage(self, st, images, thetas, noise):
        n = st["n"]
        host = torch.empty(n, 12, pin_memory=True)                # recycled by the host allocator only after the copy has run
        th = thetas.detach().to("cpu", torch.float32).reshape(n, 6)
        host[:, :6] = th
        host[:, 6:] = torch.tensor([ops.invert_affine(th[i].tolist()) for i in range(n)], dtype=torch.float32)
        st["theta"].copy_(host[:, :6].reshape(n, 2, 3), non_blocking=True)
        st["theta_inv"].copy_(host[:, 6:], non_blocking=True)
        st["images"].copy_(images, non_blocking=True)
        if noise is None:
            st["noise"].normal_()                                 # the draw `find_pred_noise` makes in the eager step (same generator, same shape)
        else:
            st["noise"].copy_(noise, non_blocking=True)

    def __call__(self, images, noise=None, thetas=None):
        """-> detached (total, equiv, sharp) like `group_step` (views of a static buffer: read them before the next call)."""
        n = int(images.shape[0])
        st = self.state.setdefault(n, {"n": n, "calls": 0})
        if st != "eager" and st.get("graph") is None and st["calls"] >= self.warmup:
            import warnings
            if not getattr(self.controller, "fused_route", False):
                if self.warn_route:
                    warnings.warn(f"GraphedStep: groups of {n} images stay on eager launches (the step does not take the fused map + "
                                  "losses node at these shapes)")
                self.state[n] = st = "eager"
            else:
                try:
                    self._capture(st, images)
                except Exception as e:                           # noqa: BLE001 -- capture is an optimisation of the host side only
                    warnings.warn(f"GraphedStep: groups of {n} images stay on eager launches (capture failed: {e})")
                    self.controller.reset()
                    self.state[n] = st = "eager"
        if st == "eager" or st.get("graph") is None:
            if st != "eager":
                st["calls"] += 1
            return group_step(self.ldm, images, self.context, self.args, self.controller, self.transform, self.denom,
                              noise=noise, thetas=thetas)
        if self.context.grad is not st["grad"]:                   # somebody dropped / replaced the gradient: the graph adds into ITS tensor
            st["grad"].zero_()
            self.context.grad = st["grad"]
        if thetas is None:
            thetas = self.transform.sample_theta(n)               # the host draws of the eager step, same order
        self.transform.last_theta_host = thetas
        self._stage(st, images, thetas, noise)
        st["graph"].replay()
        out = st["out"]
        return out[0], out[1], out[2]


def optimize_embedding(ldm, args, controllers, num_gpus, context=None,
                       from_where=["down_cross", "mid_cross", "up_cross"], draws=None, trajectory_out=None,
                       step_callback=None):
    """Reference signature (optimize.py:269-276).  `num_gpus` = devices driven by THIS process (1); the
    data-parallel width is `num_gpus * world_size`.  Returns the detached embedding [1,T,768].
    `draws = (order, noise, thetas)` injects THIS rank's image order [steps*accum], the noise of every forward in the
    reference's draw order ([2*steps*accum,4,h,w]: image view, warped view, next image ...) and the affine matrices
    [steps*accum,2,3] (parity tests; the reference takes all three from the global RNGs, optimize.py:333-365);
    `trajectory_out`, a list, receives the embedding after every optimizer step; `step_callback(step)` runs after every
    optimizer step (tools/protocol_bench.py reads its clocks there).  Images come through `GroupLoader`: the reference's
    `DataLoader` iteration (optimize.py:333-347) as a one-group-ahead host pipeline (`args.loader_workers`, 0 = the
    synchronous loop); the image ORDER is the same either way."""
    world, rank = skp_dist.world_size(), skp_dist.rank()
    width = num_gpus * world
    if args.batch_size < width or args.batch_size % width:
        raise ValueError(f"batch_size ({args.batch_size}) must be a positive multiple of the data-parallel "
                         f"width ({width}) -- the reference divides by batch_size//num_gpus (optimize.py:339)")
    accum = args.batch_size // width                              # images per rank per optimizer step
    dev, controller = next(iter(controllers.items()))
    dataset = build_dataset(args)
    transform = RandomAffineWithInverse(degrees=args.augment_degrees, scale=args.augment_scale,
                                        translate=args.augment_translate)
    if context is None:
        context = ptp_utils.init_random_noise(args.device, num_words=args.num_tokens,
                                              dim=ldm.unet.config.get("cross_attention_dim", 768))
    context = context.to(dev)
    context.requires_grad = True
    optimizer = torch.optim.Adam([context], lr=args.lr)
    reducer = skp_dist.EmbeddingReducer(context, optimizer)
    group = max(1, min(accum, getattr(args, "images_per_forward", accum)))
    log_every = getattr(args, "log_interval", 50)
    # opt-in (off: the reference encodes both views every step): latents of the un-warped views kept per dataset index,
    # at most cache_latents_max entries (64 KB each at 512^2)
    latent_cache = {} if getattr(args, "cache_latents", False) else None
    cache_max = int(getattr(args, "cache_latents_max", 200_000))
    loader = GroupLoader(dataset, _group_indices(args, len(dataset), draws, accum, group, rank, world), dev,
                         workers=int(getattr(args, "loader_workers", 4)))
    # small groups are launch-bound on the host: their step is captured once and replayed (GraphedStep); "auto" = groups of <= 2
    # images on a GPU, without the latent cache (which decides per image what to encode)
    cap = getattr(args, "capture_step", "auto")
    graphed = None
    if dev.type == "cuda" and latent_cache is None and (cap == "on" or (cap == "auto" and group <= 2)):
        graphed = GraphedStep(ldm, context, args, controller, transform, args.batch_size, warn_route=(cap == "on"))
    start = it_start = time.time()
    try:
        for step in range(int(args.num_steps)):
            running = torch.zeros(3, device=dev)
            done = 0
            while done < accum:
                it, idx, images = next(loader)
                n = len(idx)
                inject = {}
                if draws is not None:                            # `it` = this rank's iteration counter (optimize.py:339)
                    pair = torch.as_tensor(draws[1][2 * it:2 * (it + n)]).to(dev)
                    inject = dict(noise=torch.cat([pair[0::2], pair[1::2]]), thetas=torch.as_tensor(draws[2][it:it + n]))
                if latent_cache is not None and len(latent_cache) + n > cache_max:
                    latent_cache.clear()
                if graphed is not None:
                    running += torch.stack(graphed(images, **inject))
                else:
                    running += torch.stack(group_step(ldm, images, context, args, controller, transform, args.batch_size,
                                                      latent_cache=latent_cache, ids=idx, **inject))
                done += n
            reducer.step()
            if trajectory_out is not None:
                trajectory_out.append(context.detach().clone())
            if step_callback is not None:
                step_callback(step)
            if log_every and (step + 1) % log_every == 0:
                skp_dist.allreduce_sum_(running)                 # every rank holds its share of the batch mean
            if log_every and (step + 1) % log_every == 0 and rank == 0:
                r = running.tolist()
                print(f"step {step + 1}: loss {r[0]:.6f} equivariance {r[1] * args.equivariance_attn_loss_weight:.6f} "
                      f"sharpening {r[2] * args.sharpening_loss_weight:.6f} "
                      f"({(time.time() - it_start) / log_every:.3f} s/step on rank 0)", flush=True)
                it_start = time.time()
    finally:
        loader.close()
    if rank == 0:
        print(f"optimization took {time.time() - start} seconds")
    return context.detach()


def default_args(**over):
    """The reference CLI defaults that matter on this path (main.py:22-195)."""
    a = dict(dataset_name="synthetic", dataset_loc="", max_len=-1, device="cuda:0", lr=5e-3, num_steps=500,
             num_tokens=500, feature_upsample_res=128, batch_size=4, top_k_strategy="gaussian",
             furthest_point_num_samples=25, top_k=10, num_subjects=1, sharpening_loss_weight=100,
             equivariance_attn_loss_weight=1000, layers=[0, 1, 2, 3], noise_level=-1, sigma=2.0,
             augment_degrees=15, augment_scale=[0.8, 1.0], augment_translate=[0.25, 0.25], wandb=False,
             model_type="sd-legacy/stable-diffusion-v1-5", seed=0, image_size=512, log_interval=50,
             cache_latents=False, cache_latents_max=200_000, loader_workers=4, capture_step="auto")
    a.update(over)
    return SimpleNamespace(**a)
