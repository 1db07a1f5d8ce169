"""Flat parameter / gradient storage of the renderer network (uc_nerf_amd/flat.py) -- host logic, CPU only.

What the reference does with the same objects: `grad_vars` (36 tensors) handed to Adam (train.py:85-92), `state_dict()` saved as the checkpoint
pair (train.py:404-414), `load_state_dict` on reload (network/models.py:253-266).  Those surfaces must not notice that the tensors now share
one buffer.  The GPU side (the backward writing the flat gradient) is in tests/test_hip_round4.py; here a stand-in backward hands out the views.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_parallel_gloo import run_world                # noqa: E402
from uc_nerf_amd import parallel as P                          # noqa: E402
from uc_nerf_amd.flat import FlatAdam, FlatStore               # noqa: E402
from uc_nerf_amd.network.models import UCNeRF                  # noqa: E402

NO_GRAD = ("pts_bias_confidence_1.", "feature_linear_1.", "confi_linear.")


def _net(seed=0, view_num=7):
    torch.manual_seed(seed)
    return UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=24 + 12 * (view_num - 1) + 1, view_num=view_num)


def _mask(net):
    return [not any(t in n for t in NO_GRAD) for n, _ in net.named_parameters()]


def _grad_values(st, mask, scale=1.0):
    """The flat gradient a stand-in backward 'computes': a ramp, zero in the segments of the tensors the reference never differentiates."""
    g = torch.arange(st.n, dtype=torch.float32) * (1e-5 * scale) - 0.3
    for o, n, m in zip(st.offsets, st.sizes, mask):
        if not m:
            g[o:o + n] = 0
    return g


class _FakeRender(torch.autograd.Function):
    """Stands where dropin._FusedRender stands: its backward writes ONE flat gradient vector (head of a larger zero-filled pool, as
    ops.RenderPass.backward allocates it) and returns views of it."""

    @staticmethod
    def forward(ctx, st, mask, scale, *params):
        ctx.st, ctx.mask, ctx.scale = st, mask, scale
        return params[0].sum() * 0

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        pool = torch.zeros(st.grad_room + 1000)              # (+ "source gradients" behind it: a bucket must never write there)
        pool[st.grad_room:] = 7.0
        gf = pool[:st.grad_room]
        gf[:st.n] = _grad_values(st, ctx.mask, ctx.scale)
        ctx.st._test_pool = pool
        return (None, None, None) + tuple(st.grad_views(gf, ctx.mask))


def test_names_shapes_and_state_dict_round_trip_are_unchanged(sd_v7, tmp_path):
    net = _net()
    st = FlatStore.of(net)
    assert st.n == 181642 and st.is_flat() and FlatStore.of(net.nerf) is st
    names = [k for k, _ in net.named_parameters()]
    want = {k: sd_v7[k] for k in names}                       # the golden file holds the reference-initialised tensors by name
    assert sorted(names) == sorted(sd_v7) and list(net.state_dict()) == names
    assert all(tuple(net.state_dict()[k].shape) == tuple(v.shape) for k, v in want.items())
    net.load_state_dict(sd_v7)                                # copied INTO the flat buffer
    assert st.is_flat() and torch.equal(st.flat, torch.cat([v.reshape(-1) for v in want.values()]))
    # the reference's checkpoint pair (train.py:404-414) written from the flat-backed module and loaded into a fresh one
    from uc_nerf_amd.data.formats import load_checkpoint, save_checkpoint
    path = str(tmp_path / "ckpts" / "latest.tar")
    save_checkpoint(path, net, {})
    assert os.path.getsize(path) < 181642 * 4 + 65536         # the 36 views share ONE storage in the file too
    other = _net(seed=3)
    load_checkpoint(path, network_fn=other, network_mvs=None)
    assert FlatStore.of(other).is_flat() and torch.equal(FlatStore.of(other).flat, st.flat)
    ck = torch.load(path, map_location="cpu")
    assert list(ck) == ["network_fn_state_dict", "network_mvs_state_dict"] and list(ck["network_fn_state_dict"]) == names


def test_store_follows_conversions_and_repointed_parameters():
    net = _net()
    st = FlatStore.of(net)
    before = st.flat.clone()
    net.double()
    assert st.is_flat() and st.flat.dtype == torch.float64
    net.float()
    assert st.is_flat() and st.flat.dtype == torch.float32 and torch.equal(st.flat, before)
    p = net.nerf.rgb_linear.bias
    p.data = torch.full_like(p, 2.5)                          # re-pointed behind the store's back
    assert not st.is_flat()
    flat = st.sync()
    o = st.offsets[[q is p for q in st.params].index(True)]
    assert st.is_flat() and torch.equal(flat[o:o + 3], torch.full((3,), 2.5))
    with torch.no_grad():                                     # in-place writes go straight to the flat buffer (weights_init style, models.py:15-17)
        net.nerf.rgb_linear.bias.data.zero_()
    assert torch.count_nonzero(st.flat[o:o + 3]) == 0
    import copy
    twin = copy.deepcopy(net)
    ts = FlatStore.of(twin)
    ts.sync()
    assert ts is not st and ts.is_flat() and torch.equal(ts.flat, st.flat) and ts.flat.data_ptr() != st.flat.data_ptr()


def test_gradients_are_views_of_one_vector_and_untouched_tensors_keep_none():
    net = _net()
    st, mask = FlatStore.of(net), _mask(net)
    _FakeRender.apply(st, mask, 1.0, *net.parameters()).backward()
    g = st.flat_grad()
    assert g is not None and g.numel() == st.grad_room and g.data_ptr() == st._test_pool.data_ptr()
    for (name, p), o, m in zip(net.named_parameters(), st.offsets, mask):
        if m:
            assert p.grad.data_ptr() == g.data_ptr() + 4 * o, name          # installed without a copy
        else:
            assert p.grad is None, name                                    # as in the reference
    assert torch.equal(g[:st.n], _grad_values(st, mask))
    # a second backward with the gradients in place: autograd accumulates as always (and the result is still one vector: the first)
    _FakeRender.apply(st, mask, 2.0, *net.parameters()).backward()
    assert torch.allclose(torch.cat([p.grad.reshape(-1) for p, m in zip(st.params, mask) if m]),
                          torch.cat([(_grad_values(st, mask, 1.0) + _grad_values(st, mask, 2.0))[o:o + n] for o, n, m in zip(st.offsets, st.sizes, mask) if m]))
    # gradients from anywhere else are not mistaken for a flat vector
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    assert st.flat_grad() is None


def test_bucket_reduces_the_gradient_vector_in_place_with_scalars_and_flags_in_its_tail():
    net = _net()
    st, mask = FlatStore.of(net), _mask(net)
    bucket = P.FlatGradBucket(list(net.parameters()), n_scalars=2)
    assert bucket.numel == 181642 + 2 + 36
    _FakeRender.apply(st, mask, 1.0, *net.parameters()).backward()
    ptrs = [None if p.grad is None else p.grad.data_ptr() for p in net.parameters()]
    red = bucket.allreduce(0.5, [torch.tensor(1.5), 2.5])
    assert bucket.last_path == "in_place"
    assert [None if p.grad is None else p.grad.data_ptr() for p in net.parameters()] == ptrs          # nothing re-allocated, nothing copied back
    assert red.tolist() == [0.75, 1.25]
    assert torch.equal(st.flat_grad()[:st.n], _grad_values(st, mask) * 0.5)
    assert torch.all(st._test_pool[st.grad_room:] == 7.0)                                            # what lies behind the vector is untouched
    assert bucket.has_grad == mask
    # generic path (gradients that are separate tensors): same numbers
    for p, o, n, m in zip(st.params, st.offsets, st.sizes, mask):
        p.grad = _grad_values(st, mask)[o:o + n].view(p.shape).clone() if m else None
    red = bucket.allreduce(0.5, [torch.tensor(1.5), 2.5])
    assert bucket.last_path == "generic" and red.tolist() == [0.75, 1.25]
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p, m in zip(st.params, mask) if m]),
                       torch.cat([(_grad_values(st, mask) * 0.5)[o:o + n] for o, n, m in zip(st.offsets, st.sizes, mask) if m]))


def test_bucket_with_parameters_of_another_module_packs_only_those():
    net = _net()
    st, mask = FlatStore.of(net), _mask(net)
    extra = torch.nn.Linear(5, 3)                             # stands for the consistency learner's parameters (network/models.py:249-250)
    bucket = P.FlatGradBucket(list(net.parameters()) + list(extra.parameters()), n_scalars=1)
    assert bucket.numel == 181642 + 18 + 1 + 38 and st.grad_room >= bucket.numel
    _FakeRender.apply(st, mask, 1.0, *net.parameters()).backward()
    extra.weight.grad, extra.bias.grad = torch.full((3, 5), 2.0), torch.full((3,), 4.0)
    red = bucket.allreduce(0.25, [8.0])
    assert bucket.last_path == "in_place" and red.tolist() == [2.0]
    assert torch.equal(extra.weight.grad, torch.full((3, 5), 0.5)) and torch.equal(extra.bias.grad, torch.full((3,), 1.0))
    assert torch.equal(st.flat_grad()[:st.n], _grad_values(st, mask) * 0.25)


def test_flat_adam_steps_like_adam_over_the_parameter_list():
    net, ref = _net(), _net(seed=1)
    ref.load_state_dict(net.state_dict())
    st, mask = FlatStore.of(net), _mask(net)
    opt, opt_ref = FlatAdam(net, lr=5e-4, betas=(0.9, 0.999)), torch.optim.Adam(list(ref.parameters()), lr=5e-4, betas=(0.9, 0.999))     # train.py:85-92
    for it in range(3):
        opt.zero_grad()
        opt_ref.zero_grad()
        _FakeRender.apply(st, mask, 1.0 + it, *net.parameters()).backward()
        for q, o, n, m in zip(ref.parameters(), st.offsets, st.sizes, mask):
            q.grad = _grad_values(st, mask, 1.0 + it)[o:o + n].view(q.shape).clone() if m else None
        opt.step()
        opt_ref.step()
        for (name, a), b in zip(net.named_parameters(), ref.parameters()):
            torch.testing.assert_close(a, b, atol=1e-7, rtol=1e-6, msg=lambda s_: name + ": " + s_)
    assert st.is_flat()


def test_bucket_over_a_partly_frozen_network_and_flat_adam_state_round_trip(tmp_path):
    """Layers frozen by the caller (requires_grad False) drop out of the bucket but keep their place in the flat vector; FlatAdam's state
    survives state_dict() / load_state_dict() like any torch optimizer's."""
    net = _net()
    st, mask = FlatStore.of(net), _mask(net)
    for name, p in net.named_parameters():
        if name.startswith("nerf.pts_linears.0.") or name.startswith("nerf.rgb_linear."):
            p.requires_grad_(False)
    live = [m and p.requires_grad for m, p in zip(mask, net.parameters())]
    bucket = P.FlatGradBucket(list(net.parameters()), n_scalars=1)
    assert len(bucket.params) == 32 and bucket.numel == st.n + 1 + 32          # the frozen tensors' segments stay inside the reduced vector (zeros)

    class Render(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *params):
            return params[2].sum() * 0

        @staticmethod
        def backward(ctx, g):
            pool = torch.zeros(st.grad_room + 8)
            gf = pool[:st.grad_room]
            gf[:st.n] = _grad_values(st, live)
            return tuple(st.grad_views(gf, [w and n for w, n in zip(live, ctx.needs_input_grad)]))

    Render.apply(*net.parameters()).backward()
    red = bucket.allreduce(0.5, [3.0])
    assert bucket.last_path == "in_place" and red.tolist() == [1.5]
    for (name, p), o, n, w in zip(net.named_parameters(), st.offsets, st.sizes, live):
        if w:
            assert torch.equal(p.grad.reshape(-1), (_grad_values(st, live) * 0.5)[o:o + n]), name
        else:
            assert p.grad is None, name
    # optimizer state round trip
    net2 = _net(seed=2)
    net2.load_state_dict(net.state_dict())
    for p in net.parameters():
        p.requires_grad_(True)
    opt = FlatAdam(net, lr=1e-3)
    _FakeRender.apply(st, mask, 1.0, *net.parameters()).backward()
    opt.step()
    path = str(tmp_path / "opt.pt")
    torch.save(opt.state_dict(), path)
    st2 = FlatStore.of(net2)
    net2.load_state_dict(net.state_dict())
    opt2 = FlatAdam(net2, lr=1e-3)
    opt2.load_state_dict(torch.load(path))
    for o_, n_ in ((opt, net), (opt2, net2)):
        o_.zero_grad()
        s_ = FlatStore.of(n_)
        _FakeRender.apply(s_, mask, 2.0, *n_.parameters()).backward()
        o_.step()
    assert torch.equal(st.flat, st2.flat)


# ---- world size 2 over gloo: the in-place route through a real collective, one rank with an empty shard in the second step
def _two_rank_in_place(rank, world):
    net = _net()
    st, mask = FlatStore.of(net), _mask(net)
    bucket = P.FlatGradBucket(list(net.parameters()), n_scalars=1)
    out = []
    for step, scales in enumerate(((1.0, 3.0), (2.0, None))):
        for p in net.parameters():
            p.grad = None
        mine = scales[rank]
        if mine is not None:
            _FakeRender.apply(st, mask, mine, *net.parameters()).backward()
        red = bucket.allreduce(0.5, [float(rank + 1)])
        out.append((bucket.last_path, red.clone(), torch.cat([torch.zeros(n) if p.grad is None else p.grad.reshape(-1).clone() for p, n in zip(st.params, st.sizes)]),
                    [p.grad is None for p in st.params]))
    return out


def test_two_ranks_reduce_the_vector_in_place_and_an_empty_shard_still_gets_the_gradients():
    net = _net()
    st, mask = FlatStore.of(net), _mask(net)
    res = run_world(_two_rank_in_place, 2)
    for rank, steps in enumerate(res):
        path0, red0, g0, none0 = steps[0]
        assert path0 == "in_place" and red0.tolist() == [1.5]
        torch.testing.assert_close(g0, 0.5 * (_grad_values(st, mask, 1.0) + _grad_values(st, mask, 3.0)))
        assert none0 == [not m for m in mask]
        path1, red1, g1, none1 = steps[1]
        assert path1 == ("in_place" if rank == 0 else "generic")         # rank 1 differentiated nothing in step 2
        torch.testing.assert_close(g1, 0.5 * _grad_values(st, mask, 2.0))
        assert none1 == [not m for m in mask]                            # ... and still holds every gradient the other rank produced
    assert dist.is_available()


def test_a_reflattened_buffer_is_a_new_generation_whatever_its_address():
    """The "versions" weight cache (dropin.FusedSession.packed) keys on FlatStore.generation: every re-flattening counts up, because neither the new
    buffer's address (the allocator may hand the freed one out again) nor its version counter (36 copies after every flatten) tells the buffers apart."""
    net = _net()
    st = FlatStore.of(net)
    g0, v0 = st.generation, st.flat._version
    net.double()
    net.float()                                   # re-points every p.data: UCNeRF._apply re-flattens each time
    st.sync()
    g1 = st.generation
    assert g1 > g0 and st.flat._version == v0 and st.flat.dtype == torch.float32
    st.sync()
    assert st.generation == g1                   # nothing moved: the same buffer, the same generation
