"""World-size 2 / 3 gloo tests (CPU tensors) of parallel.GradReducer, the data-parallel gradient
averaging of the training path: bucket layout in reverse registration order, the first (discovery)
iteration, the overlapped iterations driven by post-accumulate hooks, parameters that never get a
gradient, a gradient that is missing in one iteration, and p.grad re-pointed at the reduced bucket.
The bucket packer is injected (torch copy); the product default is the HIP zs_copy_multi kernel."""
import ctypes
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zeroshape_amd import parallel


def torch_pack(entries, scale, device):
    """Test stand-in for hip_pack: same contract on host pointers."""
    for dst, src, n in entries:
        d = (ctypes.c_float * n).from_address(dst)
        s = (ctypes.c_float * n).from_address(src)
        torch.frombuffer(d, dtype=torch.float32).copy_(torch.frombuffer(s, dtype=torch.float32) * scale)


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(40, 300)
        self.b = torch.nn.Linear(300, 300)
        self.unused = torch.nn.Linear(7, 7)          # like the ViT's norm / head: never in the graph
        self.c = torch.nn.Linear(300, 5)
        self.sometimes = torch.nn.Parameter(torch.ones(5))

    def forward(self, x, use_sometimes=True):
        y = self.c(torch.relu(self.b(torch.relu(self.a(x)))))
        return y * self.sometimes if use_sometimes else y


def _worker(rank, world, initfile):
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)                       # replicas start DIFFERENT: the reducer must fix that
        net = Net()
        net.register_buffer("stat", torch.full((3,), float(rank)))
        red = parallel.GradReducer(net.parameters(), bucket_mb=0.3, pack_fn=torch_pack, module=net)   # several buckets
        torch.manual_seed(0)
        want = Net()                                  # rank 0's initialisation
        for (n, p), (_, q) in zip(net.named_parameters(), want.named_parameters()):
            assert torch.equal(p, q), n
        assert float(net.stat.sum()) == 0.0           # rank 0's buffer everywhere
        net.stat.fill_(float(rank) + 5)
        red.sync_buffers()
        assert float(net.stat.mean()) == 5.0
        ref = Net()
        ref.load_state_dict(net.state_dict(), strict=False)
        data = [torch.randn(world, 6, 40, generator=torch.Generator().manual_seed(100 + it)) for it in range(4)]
        for it in range(4):
            use = it != 2                             # iteration 2: `sometimes` gets no gradient
            net.zero_grad(set_to_none=True)
            net(data[it][rank], use).pow(2).mean().backward()
            red.finish()
            ref.zero_grad(set_to_none=True)
            for r in range(world):                    # the average over the ranks, computed locally
                (ref(data[it][r], use).pow(2).mean() / world).backward()
            for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                if n.startswith("unused"):
                    assert p.grad is None and q.grad is None
                elif n == "sometimes" and not use:
                    assert float(p.grad.abs().max()) == 0 and q.grad is None
                else:
                    assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), (it, n)
            if it == 0:
                assert len(red.buckets) >= 3
                assert red.buckets[0][0] is net.c.bias             # reverse registration order (root parameters come first)
                assert all(p is not net.unused.weight for b in red.buckets for p in b)
            # gradients live in the flat buckets (no copy back)
            bi, off = red.slot[id(net.b.weight)]
            assert net.b.weight.grad.data_ptr() == red.flat[bi].data_ptr() + 4 * off
        # gradient accumulation: two quiet micro-steps, the third is reduced (sum of the three, averaged)
        net.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        for k in range(3):
            red.armed = k == 2
            net(data[k][rank]).pow(2).mean().backward()
            for r in range(world):
                (ref(data[k][r]).pow(2).mean() / world).backward()
        red.finish()
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            if not n.startswith("unused"):
                assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), ("accum", n)
        red.close()
    finally:
        dist.destroy_process_group()


def _worker_uneven(rank, world, initfile):
    """First iteration: only rank 0's graph reaches `sometimes`.  The used-parameter mask is reduced
    over the ranks, so both build the same buckets (and rank 1 contributes zeros) instead of
    mismatching collective sizes; later iterations complete their buckets in different orders."""
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        net = Net()
        red = parallel.GradReducer(net.parameters(), bucket_mb=0.05, pack_fn=torch_pack, module=net)
        x = torch.randn(6, 40, generator=torch.Generator().manual_seed(50 + rank))
        for it in range(3):
            net.zero_grad(set_to_none=True)
            use = rank == 0 if it == 0 else (rank + it) % 2 == 0
            net(x, use).pow(2).mean().backward()
            red.finish()
            assert net.sometimes.grad is not None
            g = [torch.zeros_like(net.c.weight.grad) for _ in range(world)]
            dist.all_gather(g, net.c.weight.grad.contiguous())
            s = [torch.zeros_like(net.sometimes.grad) for _ in range(world)]
            dist.all_gather(s, net.sometimes.grad.contiguous())
            assert all(torch.equal(g[0], t) for t in g) and all(torch.equal(s[0], t) for t in s)
        layout = torch.tensor([len(b) for b in red.buckets])
        lay = [torch.zeros_like(layout) for _ in range(world)]
        dist.all_gather(lay, layout)
        assert all(torch.equal(lay[0], t) for t in lay) and len(red.buckets) >= 3, layout
        red.close()
    finally:
        dist.destroy_process_group()


def _replay(net, statics, loss):
    """What a hipGraph replay does: the gradients of THIS rank's graph land in the fixed tensors the capture allocated
    (`statics`), whatever `.grad` points at by now."""
    names = [n for n, p in net.named_parameters() if n in statics]
    params = dict(net.named_parameters())
    for n, g in zip(names, torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)):
        statics[n].copy_(g if g is not None else torch.zeros_like(statics[n]))


def _worker_in_place(rank, world, initfile):
    """reduce_in_place(): the captured step's form - the replays keep writing the tensors the capture allocated, the hooks are
    quiet, every bucket is packed from those tensors and all-reduced, `.grad` points at the bucket slices (no copy back)."""
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = Net()
        red = parallel.GradReducer(net.parameters(), bucket_mb=0.3, pack_fn=torch_pack, module=net)
        red.armed = False
        ref = Net()
        ref.load_state_dict(net.state_dict())
        data = [torch.randn(world, 6, 40, generator=torch.Generator().manual_seed(200 + it)) for it in range(3)]
        net(data[0][rank]).pow(2).mean().backward()                     # creates the gradient tensors ("the capture")
        statics = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
        for it in range(3):
            if it:
                _replay(net, statics, net(data[it][rank]).pow(2).mean())
            red.reduce_in_place()
            ref.zero_grad(set_to_none=True)
            for r in range(world):
                (ref(data[it][r]).pow(2).mean() / world).backward()
            for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                if n.startswith("unused"):
                    assert p.grad is None
                else:
                    assert red._grad_src[id(p)] is statics[n], n             # the replays' tensors are still the pack's source ...
                    assert p.grad.data_ptr() == red._slice(p).data_ptr() != statics[n].data_ptr(), n   # ... .grad is the bucket slice
                    assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), (it, n)
        assert len(red.buckets) >= 3 and not red.works
        # a re-capture (the engine drops the gradients first) allocates new tensors: the reducer follows them
        for p in net.parameters():
            p.grad = None
        net(data[1][rank]).pow(2).mean().backward()
        fresh = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
        red.reduce_in_place()
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            if not n.startswith("unused"):
                assert red._grad_src[id(p)] is fresh[n] and fresh[n] is not statics[n]
        ref.zero_grad(set_to_none=True)
        for r in range(world):
            (ref(data[1][r]).pow(2).mean() / world).backward()
        assert torch.allclose(net.b.weight.grad, ref.b.weight.grad, atol=1e-6, rtol=1e-5)
        red.close()
    finally:
        dist.destroy_process_group()


def _worker_in_place_uneven(rank, world, initfile):
    """reduce_in_place() when a parameter has a gradient on ONE rank only: it is in the common layout, the rank
    without a local gradient receives the average too - the ranks stay in step."""
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = Net()
        red = parallel.GradReducer(net.parameters(), bucket_mb=0.3, pack_fn=torch_pack, module=net)
        red.armed = False
        ref = Net()
        ref.load_state_dict(net.state_dict())
        xs = [[torch.randn(6, 40, generator=torch.Generator().manual_seed(70 + 10 * it + r)) for r in range(world)] for it in range(3)]
        net(xs[0][rank], use_sometimes=rank == 0).pow(2).mean().backward()       # "the capture": rank 1's graph has no `sometimes`
        assert (net.sometimes.grad is None) == (rank == 1)
        statics = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
        for it in range(3):
            # CONSECUTIVE steps: a replay overwrites the gradients of ITS graph only - the average the previous
            # reduce_in_place() gave a parameter without a local gradient must not be packed as this rank's contribution
            if it:
                _replay(net, statics, net(xs[it][rank], use_sometimes=rank == 0).pow(2).mean())
            red.reduce_in_place()
            assert net.sometimes.grad is not None and net.unused.weight.grad is None
            ref.zero_grad(set_to_none=True)
            for r in range(world):
                (ref(xs[it][r], use_sometimes=r == 0).pow(2).mean() / world).backward()
            for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                if p.grad is not None:
                    assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), (it, n)     # the TRUE mean, every step
            for p in (net.sometimes, net.c.weight):
                g = [torch.zeros_like(p.grad) for _ in range(world)]
                dist.all_gather(g, p.grad.contiguous())
                assert all(torch.equal(g[0], t) for t in g)
            assert float(net.sometimes.grad.abs().sum()) > 0
        red.close()
    finally:
        dist.destroy_process_group()


def _worker_in_place_segments(rank, world, initfile):
    """begin_in_place / launch_done / end_in_place: the segmented captured step's form.  The backward pass arrives in three
    segments (c + sometimes | b | a); a bucket goes out as soon as all its gradients are final - before the later segments
    exist - in bucket order, and the result is the plain average."""
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = Net()
        red = parallel.GradReducer(net.parameters(), bucket_mb=0.3, pack_fn=torch_pack, module=net)
        red.armed = False
        ref = Net()
        ref.load_state_dict(net.state_dict())
        segs = [[net.c.weight, net.c.bias, net.sometimes], [net.b.weight, net.b.bias], [net.a.weight, net.a.bias]]
        statics = None
        for it in range(2):
            x = [torch.randn(6, 40, generator=torch.Generator().manual_seed(300 + 10 * it + r)) for r in range(world)]
            if statics is None:
                net(x[rank]).pow(2).mean().backward()                   # "the capture"
                statics = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
            else:
                _replay(net, statics, net(x[rank]).pow(2).mean())
            truth = {n: g.clone() for n, g in statics.items()}
            name_of = {id(q): n for n, q in net.named_parameters()}
            # segments 2 and 3 "have not replayed yet": their gradient tensors hold garbage until launch_done() names them
            for p in segs[1] + segs[2]:
                statics[name_of[id(p)]].fill_(float("nan"))
            assert red.begin_in_place()

            def expected(done):       # leading buckets made of final gradients only (a bucket that straddles a cut waits)
                ids, n = {id(p) for seg in done for p in seg}, 0
                while n < len(red.buckets) and all(id(p) in ids for p in red.buckets[n]):
                    n += 1
                return n
            n1 = red.launch_done(segs[0])
            assert n1 == expected(segs[:1]) and n1 < len(red.buckets)
            for p in segs[1]:
                statics[name_of[id(p)]].copy_(truth[name_of[id(p)]])
            n2 = red.launch_done(segs[1])
            assert n2 == expected(segs[:2]) and 1 <= n2 < len(red.buckets)      # out before the last segment exists
            assert len(red.works) == n2
            for p in segs[2]:
                statics[name_of[id(p)]].copy_(truth[name_of[id(p)]])
            n3 = red.launch_done(segs[2])
            assert n3 == len(red.buckets)
            red.end_in_place()
            ref.zero_grad(set_to_none=True)
            for r in range(world):
                (ref(x[r]).pow(2).mean() / world).backward()
            for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                if not n.startswith("unused"):
                    assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), (it, n)
        red.close()
    finally:
        dist.destroy_process_group()


def test_grad_reducer_in_place_by_backward_segments():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_in_place_segments, args=(2, os.path.join(d, "init")), nprocs=2, join=True)


def test_grad_reducer_in_place_with_a_parameter_used_on_one_rank_only():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_in_place_uneven, args=(2, os.path.join(d, "init")), nprocs=2, join=True)


def test_grad_reducer_in_place_form_of_the_captured_step():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_in_place, args=(2, os.path.join(d, "init")), nprocs=2, join=True)


def test_grad_reducer_uneven_usage_across_ranks():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_uneven, args=(2, os.path.join(d, "init")), nprocs=2, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_grad_reducer_matches_averaged_gradients(world):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "init")), nprocs=world, join=True)


def test_single_process_is_a_no_op():
    net = Net()
    red = parallel.GradReducer(net.parameters())
    net(torch.randn(3, 40)).sum().backward()
    g = net.a.weight.grad.clone()
    red.finish()
    assert torch.equal(net.a.weight.grad, g) and red.buckets is None
