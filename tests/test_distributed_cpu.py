"""CPU, world_size 2, gloo: the data-parallel plumbing (shards, gradient bucket all-reduce, loss
reduce, max-over-ranks timing) that the multi-GPU runs use with backend "nccl" (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from isegprobe_amd.core.utils import distributed as D
    assert D.init_distributed("gloo")
    assert D.get_world_size() == world and D.get_rank() == rank
    # 1) disjoint, equal-size shards covering the dataset
    shard = D.shard_indices(11)
    gathered = [None] * world
    dist.all_gather_object(gathered, shard)
    # 2) gradient bucket: same model on both ranks, rank-dependent inputs
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 1))
    for p in model[0].parameters():
        p.requires_grad = rank >= 0
    bucket = D.GradBucket(model.parameters())
    x = torch.full((2, 3, 5, 5), float(rank + 1))
    bucket.zero()
    model(x).sum().backward()
    local = bucket.flat.clone()
    work = bucket.all_reduce_mean(async_op=True)
    bucket.finish(work)
    both = [None] * world
    dist.all_gather_object(both, local)
    expected = sum(both) / world
    assert torch.allclose(bucket.flat, expected)
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in bucket.params)  # grads are views of the bucket
    # 2b) the overlapped form: the "head" slice (model[2], last layer = first gradients of backward) is all-reduced from
    # inside backward, the rest after it; same averages as the one-shot collective, early parameters first in the bucket
    ob = D.GradBucket(model.parameters(), early=list(model[2].parameters()))
    assert ob.params[0] is model[2].weight and ob.n_early == 2
    fired = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, **kw: (fired.append((t.numel(), kw.get("async_op", False))), orig(t, **kw))[1]
    ob.zero()
    ob.arm_early()
    model(x).sum().backward()
    assert fired == [(4 + 1, True)], fired           # issued by the accumulate hook of the last head parameter
    ob.check_bound()
    ob.finish_overlapped()
    dist.all_reduce = orig
    assert fired[1:] == [(3 * 4 * 9 + 4, False)]
    by_param = {id(p): p.grad.clone() for p in ob.params}
    for p, off in zip(bucket.params, bucket._offsets):
        assert torch.allclose(by_param[id(p)].reshape(-1), expected[off:off + p.numel()])
    # 2c) DDP's buffer broadcast: rank-local running statistics are replaced by rank 0's
    bn = torch.nn.BatchNorm2d(4)
    bn.running_mean.fill_(float(rank + 1)), bn.running_var.fill_(3.0 * (rank + 1))
    bn.num_batches_tracked.fill_(7 * (rank + 1))
    assert D.broadcast_buffers(bn) == 3  # running_mean, running_var and the integer num_batches_tracked (DDP sends all)
    assert torch.equal(bn.running_mean, torch.ones(4)) and torch.equal(bn.running_var, torch.full((4,), 3.0))
    assert int(bn.num_batches_tracked) == 7 and bn.num_batches_tracked.dtype == torch.int64
    # 2d) a rank whose early hook never fires (its head got no gradient) still issues the same two collectives, in order
    ob.zero()
    ob.arm_early()
    fired.clear()
    dist.all_reduce = lambda t, **kw: (fired.append((t.numel(), kw.get("async_op", False))), orig(t, **kw))[1]
    if rank == 0:
        model(x).sum().backward()          # rank 0: normal backward, early slice from inside it
    else:
        model[0](x).sum().backward()       # rank 1: the head (model[2]) is not in this graph -> hook does not fire
    ob.finish_overlapped()
    dist.all_reduce = orig
    assert [n for n, _ in fired] == [4 + 1, 3 * 4 * 9 + 4], fired
    assert ob.early_fired_in_backward == (rank == 0)
    # 3) loss reduce to rank 0 and max-over-ranks timing
    red = D.reduce_loss_dict({"a": torch.tensor(float(rank + 1)), "b": torch.tensor(10.0 * (rank + 1))})
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    D.synchronize()
    if rank == 0:
        out.put(dict(shards=gathered, red={k: float(v) for k, v in red.items()}, tmax=float(t), nbytes=bucket.nbytes()))
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = res["shards"]
    assert len(s0) == len(s1) == 6 and set(s0) | set(s1) == set(range(11))
    assert set(s0) & set(s1) <= {0}  # only the wrap-around pad may repeat
    assert res["red"] == {"a": 1.5, "b": 15.0}
    assert abs(res["tmax"] - 0.2) < 1e-12
    assert res["nbytes"] == (3 * 4 * 9 + 4 + 4 + 1) * 4


def _worker8(rank, world, port, out):
    """World size 8 with BASELINE configs[4]'s bucket: ConvSegHead(768, 2, 1) = 10 619 137 + click PatchEmbed(3 -> 768, 14) = 452 352
    trainable elements = 11.07 M (44.3 MB fp32)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from isegprobe_amd.core.utils import distributed as D
    torch.set_num_threads(1)
    assert D.init_distributed("gloo") and D.get_world_size() == world
    n_img = 256 * 3 + 5  # a dataset that does not divide by 8: every rank gets ceil(n / 8) indices, the tail wraps around
    shard = D.shard_indices(n_img)
    head = torch.nn.Parameter(torch.zeros(2 * (768 * 768 * 9 + 768) + 768 + 1))
    embed = torch.nn.Parameter(torch.zeros(3 * 14 * 14 * 768 + 768))
    bucket = D.GradBucket([embed, head], early=[head])  # given in the model's order; the head is moved to the front
    assert bucket.params[0] is head and bucket.flat.numel() == 11071489 and bucket.nbytes() == 44285956
    bucket.zero()
    bucket.arm_early()
    # "backward": d/dhead = rank + 1, d/dembed = 10 (rank + 1); the head's gradient is accumulated first
    (head.sum() * float(rank + 1) + embed.sum() * float(10 * (rank + 1))).backward()
    bucket.check_bound()
    bucket.finish_overlapped()
    mean = sum(range(1, world + 1)) / world
    ok = (bucket.early_fired_in_backward and torch.all(head.grad == mean).item() and torch.all(embed.grad == 10 * mean).item())
    # one-shot fallback gives the same averages
    bucket.zero()
    (head.sum() * float(rank + 1) + embed.sum() * float(10 * (rank + 1))).backward()
    bucket.all_reduce_mean()
    ok = ok and torch.all(head.grad == mean).item() and torch.all(embed.grad == 10 * mean).item()
    bn = torch.nn.BatchNorm2d(788)  # LoftUp(768)'s 788-channel BatchNorms
    bn.running_mean.fill_(float(rank)), bn.num_batches_tracked.fill_(rank)
    D.broadcast_buffers(bn)
    ok = ok and float(bn.running_mean.abs().max()) == 0.0 and int(bn.num_batches_tracked) == 0
    gathered = [None] * world
    dist.all_gather_object(gathered, (shard, bool(ok)))
    D.synchronize()
    if rank == 0:
        out.put(gathered)
    dist.destroy_process_group()


def test_eight_rank_gloo_configs4_bucket():
    """The first 8-GPU run must be a measurement, not a debug session: the rank-count-dependent host logic (shards, the
    overlapped gradient all-reduce with the early slice, the one-shot fallback, the buffer broadcast) at world size 8 with
    configs[4]'s 11.07 M-element bucket, on gloo."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, out)) for r in range(8)]
    for p in procs:
        p.start()
    res = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 256 * 3 + 5
    shards = [r[0] for r in res]
    assert all(r[1] for r in res)
    assert all(len(s) == (n + 7) // 8 for s in shards)
    flat = [i for s in shards for i in s]
    assert set(flat) == set(range(n)) and len(flat) - n == 8 * ((n + 7) // 8) - n  # every image once, plus the wrapped tail


def test_bench_eight_rank_train_dry_run():
    """`python bench.py --gpus 8 --mode train --dry-run`: the self-spawned 8-rank launch path of the scaling bench (rendezvous,
    barrier-bracketed region, max over ranks, ONE line from rank 0) on CPU + gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--mode", "train"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["mode"] == "train" and lines[0]["max_dt"] >= 0.08


def test_single_process_noops():
    from isegprobe_amd.core.utils import distributed as D
    assert D.get_world_size() == 1 and D.get_rank() == 0
    assert D.shard_indices(5) == [0, 1, 2, 3, 4]
    d = {"x": torch.tensor(1.0)}
    assert D.reduce_loss_dict(d) is d
    p = torch.nn.Parameter(torch.ones(3))
    b = D.GradBucket([p])
    (p * 2).sum().backward()
    assert b.all_reduce_mean() is None and torch.equal(b.flat, torch.full((3,), 2.0))


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_multi_gpu_launch_plumbing(launcher):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own two ranks (round 1 exited with SystemExit there);
    under torch.distributed.run it uses the ranks it is given.  Dry run: CPU + gloo, no GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--mode", "train"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29641"] + cmd[1:]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["max_dt"] >= 0.02  # max over ranks, printed once


def test_grad_bucket_survives_zero_grad():
    """zero_grad(set_to_none=True) -- torch's default -- detaches .grad from the flat bucket; the all-reduce would then
    average stale zeros and the replicas diverge.  check_bound() catches it, zero() re-binds the views."""
    from isegprobe_amd.core.utils import distributed as D
    m = torch.nn.Linear(4, 3)
    b = D.GradBucket(m.parameters())
    m(torch.ones(2, 4)).sum().backward()
    assert b.flat.abs().sum() > 0
    b.check_bound()
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad()  # set_to_none=True
    with pytest.raises(RuntimeError):
        b.check_bound()
    b.zero()
    b.check_bound()
    m(torch.ones(2, 4)).sum().backward()
    assert b.flat.abs().sum() > 0 and m.weight.grad.data_ptr() == b.flat.data_ptr()


class _StubDataset:
    """Seven images with 1-3 objects each (uneven over three ranks), seeded ellipses."""

    def __init__(self, n=7):
        import numpy as np
        self.np, self.n = np, n

    def __len__(self):
        return self.n

    def get_sample(self, index):
        np = self.np
        rng = np.random.default_rng(100 + index)
        yy, xx = np.mgrid[:40, :56]
        n_obj = 1 + index % 3
        masks = []
        for _ in range(n_obj):
            cy, cx, ry, rx = rng.integers(8, 32), rng.integers(8, 48), rng.integers(4, 12), rng.integers(4, 16)
            masks.append(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1)
        image = rng.integers(0, 255, (40, 56, 3)).astype(np.uint8)

        class _S:
            objects_ids = list(range(n_obj))

            def gt_mask(self, i):
                return masks[i]
        s = _S()
        s.image = image
        return s


class _StubPredictor:
    """A deterministic 'network': a probability map that depends on the image and grows a disk around every positive click."""
    device = "cpu"

    def set_input_image(self, image):
        import numpy as np
        self.base = image[..., 0].astype(np.float32) / 1024.0

    def get_prediction(self, clicker):
        import numpy as np
        p = self.base.copy()
        yy, xx = np.mgrid[:p.shape[0], :p.shape[1]]
        for k, c in enumerate(clicker.clicks_list):
            d = (yy - c.coords[0]) ** 2 + (xx - c.coords[1]) ** 2 <= (4 + k) ** 2
            p[d] = 0.9 if c.is_positive else 0.1
        return p


def _eval_worker(rank, world, port, out, n_images=7):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from isegprobe_amd.core.inference.evaluation import evaluate_dataset
    dist.init_process_group("gloo", init_method="env://")
    ious, elapsed = evaluate_dataset(_StubDataset(n_images), _StubPredictor(), shard=(rank, world), pred_thr=0.5, max_iou_thr=0.6,
                                     max_clicks=6, device_clicker=False)
    if rank == world - 1:  # every rank holds the gathered list; report the last one's
        out.put(([a.tolist() for a in ious], elapsed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images", [(3, 7), (4, 3)])
def test_sharded_evaluation_equals_single_process(world, n_images):
    """evaluate_dataset(shard=(rank, world)) on gloo ranks: the gathered per-object IoU arrays are the single-process list,
    element for element and in its order (1-3 objects per image: uneven shards; with 4 ranks and 3 images one rank has nothing)."""
    from isegprobe_amd.core.inference.evaluation import evaluate_dataset
    from isegprobe_amd.core.inference.utils import compute_noc_metric
    ref, _ = evaluate_dataset(_StubDataset(n_images), _StubPredictor(), pred_thr=0.5, max_iou_thr=0.6, max_clicks=6, device_clicker=False)
    assert len(ref) == sum(1 + i % 3 for i in range(n_images))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, out, n_images)) for r in range(world)]
    for p in procs:
        p.start()
    got, elapsed = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == len(ref) and elapsed > 0
    import numpy as np
    for a, b in zip(got, ref):
        assert np.array_equal(np.asarray(a, np.float32), b)
    assert compute_noc_metric([np.asarray(a, np.float32) for a in got], [0.8, 0.9], 6) == compute_noc_metric(ref, [0.8, 0.9], 6)
