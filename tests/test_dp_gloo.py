"""Data-parallel path on CPU: world_size 2, gloo.  Each rank runs the block on its minibatch shard (the oracle stands in
for the kernels here -- tests may use it), the flat parameter-gradient bucket is averaged by GradExchange, and the result
must equal what a single process gets on the whole batch (DDP semantics: mean over replicas)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, synth


def _worker(rank, world, port, B, C, H, W, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mga_yolo_amd.dp import GradExchange, shard_batch
    from oracle import maskcbam_oracle as O
    x, mask, gy = synth(B, C, H, W, seed=42)
    p = O.Params.default_init(C)
    sl = shard_batch(B, world, rank)
    xs, ms, gs = x[sl], mask[sl], gy[sl]
    y, ctx = O.forward(xs, ms, p)
    g = O.backward(gs, xs, ms, p, O.Config(), ctx)
    names = ("gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta")
    bucket = torch.cat([g[n].reshape(-1) for n in names])      # one flat bucket, as PyramidPlan.grad_bucket
    ex = GradExchange(bucket)
    ex.start()
    overlapped = float(y.sum())                                 # independent work between start() and finish()
    ex.finish()
    ex.finish()                                                 # idempotent when nothing is pending
    # plain arrays on the queue: torch tensors travel as shared-memory handles that die with the producer process
    q.put((rank, bucket.numpy().copy(), g["gx"].numpy().copy(), sl.start, sl.stop, overlapped))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_exchange_matches_single_process():
    from oracle import maskcbam_oracle as O
    B, C, H, W, world = 4, 32, 10, 10, 2
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx_mp.Process(target=_worker, args=(r, world, port, B, C, H, W, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = [q.get(timeout=240) for _ in range(world)]
    got = [(r, torch.from_numpy(b), torch.from_numpy(gx), lo, hi, ov) for r, b, gx, lo, hi, ov in got]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    x, mask, gy = synth(B, C, H, W, seed=42)
    p = O.Params.default_init(C)
    y, c = O.forward(x, mask, p)
    g = O.backward(gy, x, mask, p, O.Config(), c)
    want = torch.cat([g[n].reshape(-1) for n in ("gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta")]) / world
    for rank, bucket, gx, lo, hi, _ in got:
        assert (bucket - want).abs().max() <= 1e-5 * want.abs().max(), rank      # same averaged bucket on every rank
        assert (gx - g["gx"][lo:hi]).abs().max() <= 1e-6 * g["gx"].abs().max()    # no collective in the data path
    assert torch.equal(got[0][1], got[1][1])


def _ddp_worker(rank, world, port, B, C, H, W, q):
    """The module inside the wrapper the reference's trainer uses: DistributedDataParallel(find_unused_parameters=True)
    (U/engine/trainer.py:366-367), host path, with and without a mask (without one every parameter is still used)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from mga_yolo_amd import MaskCBAM
    from mga_yolo_amd.dp import shard_batch
    torch.manual_seed(0)
    m = MaskCBAM(C)
    ddp = DDP(m, find_unused_parameters=True)
    x, mask, gy = synth(B, C, H, W, seed=43)
    sl = shard_batch(B, world, rank)
    out = {}
    for tag, inp in (("mask", [x[sl], mask[sl]]), ("nomask", x[sl])):
        ddp.zero_grad()
        y = ddp(inp)
        (y * gy[sl]).sum().backward()
        out[tag] = {n: p.grad.numpy().copy() for n, p in m.named_parameters()}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_module_inside_ddp_wrapper_averages_gradients():
    from mga_yolo_amd import MaskCBAM
    B, C, H, W, world = 4, 32, 10, 10, 2
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx_mp.Process(target=_ddp_worker, args=(r, world, port, B, C, H, W, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    torch.manual_seed(0)
    m = MaskCBAM(C)
    x, mask, gy = synth(B, C, H, W, seed=43)
    for tag, inp in (("mask", [x, mask]), ("nomask", x)):
        m.zero_grad()
        (m(inp) * gy).sum().backward()
        for n, p in m.named_parameters():
            want = (p.grad / world).numpy()                          # DDP: mean over replicas of the per-shard sums
            for r in range(world):
                g = got[r][tag][n]
                assert abs(g - want).max() <= 1e-5 * max(abs(want).max(), 1e-12), (tag, n, r)
            assert (got[0][tag][n] == got[1][tag][n]).all()


def test_shard_batch_is_an_equal_contiguous_partition():
    from mga_yolo_amd.dp import shard_batch
    seen = []
    for r in range(8):
        s = shard_batch(256, 8, r)
        assert s.stop - s.start == 32
        seen += list(range(s.start, s.stop))
    assert seen == list(range(256))
    with pytest.raises(ValueError):
        shard_batch(30, 8, 0)


def test_exchange_is_a_no_op_without_a_process_group():
    from mga_yolo_amd.dp import GradExchange
    b = torch.arange(6, dtype=torch.float32)
    ex = GradExchange(b)
    ex.start(); ex.finish()
    assert ex.world == 1 and torch.equal(b, torch.arange(6, dtype=torch.float32))


def _harness_worker(rank, world, port, q):
    """SURVEY 8d images/s definition (2): the train-step harness (tools/harness.py: backbone stand-in -> MGAMaskHead -> MaskCBAM ->
    SegmentationLoss -> Kendall combine, the loss formed inside forward) inside the reference's wrapper,
    DistributedDataParallel(find_unused_parameters=True) (U/engine/trainer.py:366-367), two gloo ranks on the host."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import harness
    dev = torch.device("cpu")
    model, wrapped, opt = harness.build("n", dev, world, depth=1)
    img, masks = harness.synthetic_batch(2, 64, dev, seed=100 + rank)
    opt.zero_grad(set_to_none=True)
    loss = wrapped(img, masks)
    loss.backward()
    grads = {n: p.grad.numpy().copy() for n, p in model.named_parameters()}
    res = harness.run("n", batch=2, size=64, steps=2, warmup=1, device=dev, world=world, rank=rank)     # the timed loop runs under DDP too
    q.put((rank, float(loss), grads, res["images_per_s"], res["ddp"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_train_harness_inside_ddp_matches_the_mean_of_the_shards():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import harness
    world = 2
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = 29700 + (os.getpid() % 90)
    procs = [ctx_mp.Process(target=_harness_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = {r: (l, g, ips, ddp) for r, l, g, ips, ddp in (q.get(timeout=500) for _ in range(world))}
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # single process: the same model on each shard in turn, gradients averaged by hand
    dev = torch.device("cpu")
    want = None
    for r in range(world):
        model, wrapped, opt = harness.build("n", dev, 1, depth=1)
        img, masks = harness.synthetic_batch(2, 64, dev, seed=100 + r)
        loss = wrapped(img, masks)
        assert abs(float(loss) - got[r][0]) <= 1e-5 * abs(float(loss))
        loss.backward()
        g = {n: p.grad.clone() / world for n, p in model.named_parameters()}
        want = g if want is None else {n: want[n] + g[n] for n in g}
    names = set(want)
    assert any(n.startswith("blocks.") for n in names) and any(n.startswith("heads.") for n in names) and "mtl_log_vars" in names
    for n, w in want.items():
        for r in range(world):
            assert abs(got[r][1][n] - w.numpy()).max() <= 2e-5 * max(float(w.abs().max()), 1e-8), (n, r)
        assert (got[0][1][n] == got[1][1][n]).all(), n
    assert all(got[r][2] > 0 and got[r][3] for r in range(world))
