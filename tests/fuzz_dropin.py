"""Stateful fuzzer of the rendering() drop-in (network/renderer.py:215-255 mirror): does a long-lived network + session -- packed-weight caches,
source copies, prepared parameter structs, workspaces, the flat parameter store -- ever answer differently from a freshly built one?

    python tests/fuzz_dropin.py [--cases 12] [--steps 40] [--seed 0]

One case = one small scene and ONE network object driven through a random sequence of the things a training / evaluation script does to it:
  render      rendering() under no_grad on a random batch (ragged ray counts, 30 / 45 / 90 samples, random pixels or an image chunk)
  train       rendering() under autograd, a random loss, backward, an optimizer step (torch.optim.Adam over the parameters or flat.FlatAdam)
  delayed     a training forward whose backward runs later: parameters scaled in place, no_grad renders in every precision and another training
              forward in between -- the gradients must be those of an immediate backward
  weights     parameters scaled in place / load_state_dict / a round trip through the CPU (re-flattens the store)
  sources     the volumes / images / features written in place; replaced by new tensors; switched between channel-major and channel-last
              (channels_last_3d / channels_last) layouts or handed over as strided slices of wider allocations; only the confidence map or only
              the poses changed; (a third of the batches arrive as non-contiguous views as well)
  scene       the script turns to a second scene of another size (and later back): two sets of sources alternate through one session
  knobs       inference precision (bf16x3_fused / bf16x3 / f32), weight-cache policy (verify / versions), the one-launch tail route on / off, the
              split's terms (bf16 / fp16)
After every render the same call is made on a FRESH network (new module, state_dict copied in, new session) with fresh clones of the sources:
no_grad outputs must be bit-identical; a training call's rgb / depth bit-identical and its gradients equal up to float-atomic order
(2e-5 max|g| + 1e-4 relative; the one-element gradient of a density head's bias, a sum over all samples that cancels, 1e-3; an element outside that is
held to three times the difference between TWO fresh networks running the same step before it counts as a failure).  No oracle involved: this checks state, not arithmetic (the other fuzzers do that).
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEV = "cuda:0"


def _mods():
    import uc_nerf_amd
    uc_nerf_amd.install_dropin()
    import network.models as models
    import network.renderer as renderer
    return types.SimpleNamespace(models=models, renderer=renderer)


def _net(mods, V, sd):
    m = mods.models.UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=24 + 12 * (V - 1) + 1, skips=[4], view_num=V)
    m.load_state_dict(sd)
    return m.to(DEV)


def _layout(t, cl, kind):
    """A fresh tensor with t's values in the asked layout (kind 'vol': [1,8,D,h,w]; 'feat': [V,1,8,H,W]; 'imgs': [1,V,3,H,W])."""
    t = t.detach().clone().contiguous()
    if not cl:
        return t
    if kind == "vol":
        return t.contiguous(memory_format=torch.channels_last_3d)
    if kind == "feat":
        return t[:, 0].contiguous(memory_format=torch.channels_last).unsqueeze(1)
    return t[0].contiguous(memory_format=torch.channels_last).unsqueeze(0)


def grads_equal(x, y, m_all):
    """Equal up to float-atomic order: 2e-5 max|g| + 1e-4 relative per element; a ONE-element tensor (the bias of a density head: a sum over all samples
    that cancels, re-ordered by every launch's atomics -- two fresh networks differ by up to 2e-4 of it, more when it happens to cancel to nearly
    nothing) 1e-3 relative + 2e-5 of the step's largest gradient element `m_all`."""
    m = max(y.abs().max().item(), 1e-6)
    if y.numel() == 1:
        return bool(((x - y).abs() <= 1e-3 * m + 2e-5 * m_all).all())
    return bool(((x - y).abs() <= 2e-5 * m + 1e-4 * y.abs()).all())


def run(cases=12, steps=40, seed=0, verbose=True, only=None, magnify=False):
    import uc_nerf_amd
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd import flat as F
    from uc_nerf_amd.synthetic import cascade_outputs, init_ucnerf_state_dict, live_path_batch, make_scene, scene_to
    mods = _mods()
    e_p, _ = mods.models.get_embedder(10, 0)
    e_d, _ = mods.models.get_embedder(4, 0)
    qfn = lambda pts, vd, f, fn: mods.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d)      # noqa: E731
    failures, renders, trains = [], 0, 0
    t0 = time.time()
    for c in (range(cases) if only is None else [only]):
        s = seed * 1000 + c
        rng = np.random.RandomState(s)
        V = int(rng.choice([2, 4, 7]))
        H, W = 4 * int(rng.randint(8, 25)), 4 * int(rng.randint(8, 33))
        scene = scene_to(make_scene(seed=s, H=H, W=W, V=V, small_volumes=bool(rng.rand() < 0.5)), torch.device(DEV))
        outputs = cascade_outputs(scene, seed=s)
        args = types.SimpleNamespace(view_num=V, feat_dim=24 + 12 * (V - 1) + 1, img_downscale=1.0, use_color_volume=False, net_type="v2")
        net = _net(mods, V, init_ucnerf_state_dict(seed=s, n_src=V - 1, sigma_scale=0.05, sigma_bias=0.05))
        opt = None
        cl = [False] * 5                                  # layout of vol1..3, img_feat, imgs
        src = dict(vols=[_layout(v, False, "vol") for v in scene["vols"]], img_feat=_layout(scene["img_feat"], False, "feat"),
                   imgs=_layout(scene["imgs"], False, "imgs"), conf=scene["confidence"].clone(), w2cs=scene["w2cs"].clone(), intr=scene["intrinsics"].clone())
        knobs = dict(precision="bf16x3_fused", cache="verify", tail=1, operand="bf16")
        log, other = [], None

        def apply_knobs():
            uc_nerf_amd.set_inference_precision(knobs["precision"])
            uc_nerf_amd.set_weight_cache(knobs["cache"])
            L.lib().ucnerf_set_fused_tail(knobs["tail"])
            uc_nerf_amd.set_split_operand(knobs["operand"])

        def call(network, sources, batch, grad):
            vf = {"stage%d" % (i + 1): {"volume_feature_no_ref": sources["vols"][i]} for i in range(3)}
            pose = {"w2cs": sources["w2cs"].clone(), "intrinsics": sources["intr"].clone()}
            with torch.enable_grad() if grad else torch.no_grad():
                return mods.renderer.rendering(args, pose, batch["rays_pts"], batch["rays_ndc"], batch["depth_candidates"], batch["rays_dir"], vf,
                                               sources["imgs"], network_fn=network, img_feat=sources["img_feat"], network_query_fn=qfn,
                                               confidence=sources["conf"])

        def fresh():
            n2 = _net(mods, V, {k: v.detach().clone() for k, v in net.state_dict().items()})
            s2 = dict(vols=[_layout(v, k_, "vol") for v, k_ in zip(src["vols"], cl[:3])], img_feat=_layout(src["img_feat"], cl[3], "feat"),
                      imgs=_layout(src["imgs"], cl[4], "imgs"), conf=src["conf"].clone(), w2cs=src["w2cs"].clone(), intr=src["intr"].clone())
            return n2, s2

        def batch_of(k):
            n = int(rng.choice([1, 31, 64, 100, 257, 700, 1024]))
            S = int(rng.choice([30, 45, 90]))
            chunk = None if rng.rand() < 0.5 else int(rng.randint(0, max(1, H * W // n)))
            b = live_path_batch(scene, outputs, n, S, seed=s * 100 + k, chunk_idx=chunk)
            if rng.rand() < 0.3:
                # the same values behind other strides: slices of wider allocations, every second column, an expanded direction
                def wide(t):
                    w_ = torch.zeros(t.shape[:-1] + (t.shape[-1] + 3,), device=t.device)
                    w_[..., :t.shape[-1]] = t
                    return w_[..., :t.shape[-1]]
                z2 = torch.zeros(n, 2 * S, device=DEV)
                z2[:, ::2] = b["depth_candidates"]
                b = dict(b, rays_pts=wide(b["rays_pts"]), rays_ndc={k_: wide(v) for k_, v in b["rays_ndc"].items()}, depth_candidates=z2[:, ::2],
                         rays_dir=wide(b["rays_dir"]))
            return b

        try:
            apply_knobs()
            for k in range(steps):
                op = rng.choice(["render", "render", "render", "train", "delayed", "weights", "sources", "sources", "knobs", "scene"])
                if op == "render":
                    b = batch_of(k)
                    got = call(net, src, b, False)
                    n2, s2 = fresh()
                    want = call(n2, s2, b, False)
                    renders += 1
                    ok = torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
                    log.append("render n=%d S=%d" % tuple(b["depth_candidates"].shape))
                    if not ok:
                        raise AssertionError("render differs from a fresh network's: max |d rgb| %.3g" % float((got[0] - want[0]).abs().max()))
                elif op == "train":
                    b = batch_of(k)
                    kind = int(rng.randint(0, 3))
                    if opt is None or rng.rand() < 0.3:
                        opt = (torch.optim.Adam(net.parameters(), lr=1e-3), F.FlatAdam(net, lr=1e-3), None)[kind]
                    n2, s2 = fresh()
                    need = bool(rng.rand() < 0.5)                    # gradients into the sources as well
                    outs = []
                    for network, sources in ((net, src), (n2, s2)):
                        leaves = []
                        if need:
                            sources = dict(sources, vols=[v.detach().requires_grad_(True) for v in sources["vols"]],
                                           img_feat=sources["img_feat"].detach().requires_grad_(True))
                            leaves = sources["vols"] + [sources["img_feat"]]
                        for p in network.parameters():
                            p.grad = None
                        rgb, depth = call(network, sources, b, True)
                        gen = torch.Generator().manual_seed(s * 100 + k)
                        r3 = torch.randn(rgb.shape, generator=gen).to(DEV)
                        r1 = torch.randn(depth.shape, generator=gen).to(DEV)
                        ((rgb * r3).sum() + (depth * r1).sum()).backward()
                        outs.append((rgb.detach(), depth.detach(), [p.grad for p in network.parameters()] + [t.grad for t in leaves]))
                    trains += 1
                    log.append("train n=%d S=%d sources_need_grad=%d" % (tuple(b["depth_candidates"].shape) + (need,)))
                    (rgb_a, d_a, g_a), (rgb_b, d_b, g_b) = outs
                    if not (torch.equal(rgb_a, rgb_b) and torch.equal(d_a, d_b)):
                        raise AssertionError("training forward differs from a fresh network's")
                    m_all = max(float(y.abs().max()) for y in g_b if y is not None)
                    for gi, (x, y) in enumerate(zip(g_a, g_b)):
                        if (x is None) != (y is None):
                            raise AssertionError("gradient present on one side only")
                        if x is not None:
                            m = max(y.abs().max().item(), 1e-6)
                            if not grads_equal(x, y, m_all):
                                # the floor: the same step on a SECOND fresh network (float atomics add in a different order every launch)
                                n3, s3 = fresh()
                                if need:
                                    s3 = dict(s3, vols=[v.detach().requires_grad_(True) for v in s3["vols"]], img_feat=s3["img_feat"].detach().requires_grad_(True))
                                rgb3, depth3 = call(n3, s3, b, True)
                                ((rgb3 * r3).sum() + (depth3 * r1).sum()).backward()
                                g3 = ([p.grad for p in n3.parameters()] + ([t.grad for t in s3["vols"] + [s3["img_feat"]]] if need else []))[gi]
                                floor = float((g3 - y).abs().max())
                                if float((x - y).abs().max()) > 3 * floor + 2e-5 * m:
                                    raise AssertionError("gradient %d of %d (%s) differs from a fresh network's: %.3g of max|g| (two fresh networks differ by %.3g)" % (
                                        gi, len(g_a), tuple(y.shape), float((x - y).abs().max()) / m, floor / m))
                    if opt is not None:
                        opt.step()
                elif op == "delayed":
                    # a training forward whose backward runs LATER: in between the parameters are scaled in place and other renders come by (the same
                    # precision as the training forward among them); the backward must still differentiate the forward that was run
                    b = batch_of(k)
                    n2, s2 = fresh()
                    outs = []
                    for network, sources, disturb in ((net, src, True), (n2, s2, False)):
                        for p in network.parameters():
                            p.grad = None
                        rgb, depth = call(network, sources, b, True)
                        gen = torch.Generator().manual_seed(s * 100 + k)
                        loss = (rgb * torch.randn(rgb.shape, generator=gen).to(DEV)).sum() + (depth * torch.randn(depth.shape, generator=gen).to(DEV)).sum()
                        if disturb:
                            if rng.rand() < 0.7:
                                with torch.no_grad():
                                    for p in network.parameters():
                                        p.mul_(1.02)
                            for _ in range(int(rng.randint(1, 3))):
                                uc_nerf_amd.set_inference_precision(str(rng.choice(["f32", "bf16x3_fused", "bf16x3"])))
                                call(network, sources, batch_of(k + 1000), False)
                            if rng.rand() < 0.3:                       # another training forward over the same pass (overwrites the kept activations)
                                call(network, sources, batch_of(k + 2000) if rng.rand() < 0.5 else b, True)
                            apply_knobs()
                        loss.backward()
                        outs.append([p.grad for p in network.parameters()])
                    trains += 1
                    log.append("delayed backward n=%d S=%d" % tuple(b["depth_candidates"].shape))
                    third = None
                    m_all = max(float(y.abs().max()) for y in outs[1] if y is not None)
                    for gi, (x, y) in enumerate(zip(*outs)):
                        if (x is None) != (y is None):
                            raise AssertionError("delayed backward: gradient present on one side only")
                        if x is not None:
                            m = max(y.abs().max().item(), 1e-6)
                            if not grads_equal(x, y, m_all):
                                if third is None:                      # the floor: a second immediate backward on a network in the forward-time state
                                    for p in n2.parameters():
                                        p.grad = None
                                    rgb3, depth3 = call(n2, s2, b, True)
                                    gen = torch.Generator().manual_seed(s * 100 + k)
                                    ((rgb3 * torch.randn(rgb3.shape, generator=gen).to(DEV)).sum() + (depth3 * torch.randn(depth3.shape, generator=gen).to(DEV)).sum()).backward()
                                    third = [p.grad for p in n2.parameters()]
                                floor = float((third[gi] - y).abs().max())
                                if float((x - y).abs().max()) > 3 * floor + 2e-5 * m:
                                    raise AssertionError("delayed backward: gradient %d (%s) differs from an immediate one: %.3g of max|g| (two immediate ones differ by %.3g)" % (
                                        gi, tuple(y.shape), float((x - y).abs().max()) / m, floor / m))
                    opt = None                                         # (the parameters may have been scaled: drop optimizer state like a script would)
                elif op == "weights":
                    how = int(rng.randint(0, 3))
                    if how == 0:
                        with torch.no_grad():
                            for p in net.parameters():
                                p.mul_(1.0 + 0.01 * float(rng.randn()))
                    elif how == 1:
                        net.load_state_dict({k_: v * 1.01 for k_, v in net.state_dict().items()})
                    else:
                        net.cpu()
                        net.to(DEV)
                        opt = None                                 # (optimizer state points at the old parameter storage; a script would rebuild it too)
                    log.append("weights how=%d" % how)
                elif op == "sources":
                    how = int(rng.randint(0, 6))
                    if how == 5:                                   # neither channel-major nor channel-last: slices of wider allocations (copied by the library)
                        def sliced(t):
                            w_ = torch.zeros(t.shape[:-1] + (t.shape[-1] + 2,), device=t.device)
                            w_[..., 1:-1] = t
                            return w_[..., 1:-1]
                        src["vols"] = [sliced(v.contiguous()) for v in src["vols"]]
                        src["img_feat"], src["imgs"], src["conf"] = sliced(src["img_feat"].contiguous()), sliced(src["imgs"].contiguous()), sliced(src["conf"])
                        cl = [False] * 5
                    elif how == 0:                                 # written in place
                        i = int(rng.randint(0, 5))
                        t = (src["vols"] + [src["img_feat"], src["imgs"]])[i]
                        with torch.no_grad():
                            t.mul_(0.9)
                    elif how == 1:                                 # replaced by new tensors of the same layout
                        src["vols"] = [_layout(v * 1.1, k_, "vol") for v, k_ in zip(src["vols"], cl[:3])]
                        src["img_feat"] = _layout(src["img_feat"] * 1.1, cl[3], "feat")
                    elif how == 2:                                 # layouts switched
                        cl = [bool(rng.rand() < 0.5) for _ in range(5)]
                        src["vols"] = [_layout(v, k_, "vol") for v, k_ in zip(src["vols"], cl[:3])]
                        src["img_feat"], src["imgs"] = _layout(src["img_feat"], cl[3], "feat"), _layout(src["imgs"], cl[4], "imgs")
                    elif how == 3:                                 # the light sources only
                        with torch.no_grad():
                            src["conf"].mul_(0.95)
                    else:
                        src["w2cs"] = src["w2cs"].clone()
                        src["w2cs"][1:, 0, 3] += 0.001
                    log.append("sources how=%d cl=%s" % (how, "".join(str(int(x)) for x in cl)))
                elif op == "scene":
                    # the script turns to its OTHER scene (a validation view of another size, say) and later back: the sources of the one it leaves
                    # are neither touched nor freed
                    if other is None:
                        H2, W2 = 4 * int(rng.randint(8, 25)), 4 * int(rng.randint(8, 33))
                        sc2 = scene_to(make_scene(seed=s + 500, H=H2, W=W2, V=V, small_volumes=bool(rng.rand() < 0.5)), torch.device(DEV))
                        cl2 = [bool(rng.rand() < 0.5) for _ in range(5)]
                        src2 = dict(vols=[_layout(v, k_, "vol") for v, k_ in zip(sc2["vols"], cl2[:3])], img_feat=_layout(sc2["img_feat"], cl2[3], "feat"),
                                    imgs=_layout(sc2["imgs"], cl2[4], "imgs"), conf=sc2["confidence"].clone(), w2cs=sc2["w2cs"].clone(), intr=sc2["intrinsics"].clone())
                        other = (sc2, cascade_outputs(sc2, seed=s + 500), src2, cl2, H2, W2)
                    (scene, outputs, src, cl, H, W), other = other, (scene, outputs, src, cl, H, W)
                    log.append("scene -> %dx%d cl=%s" % (H, W, "".join(str(int(x)) for x in cl)))
                elif op == "knobs":
                    knobs["precision"] = str(rng.choice(["bf16x3_fused", "bf16x3_fused", "bf16x3", "f32"]))
                    knobs["cache"] = str(rng.choice(["verify", "versions"]))
                    knobs["tail"] = int(rng.rand() < 0.7)
                    knobs["operand"] = str(rng.choice(["bf16", "bf16", "fp16"]))
                    apply_knobs()
                    log.append("knobs %s" % knobs)
                if magnify and op != "render":
                    # (one case under the magnifying glass: a render check after EVERY step, on a fixed batch that does not consume the case's draws)
                    bchk = live_path_batch(scene, outputs, 64, 45, seed=4242, chunk_idx=None)
                    got = call(net, src, bchk, False)
                    n2, s2 = fresh()
                    want = call(n2, s2, bchk, False)
                    if not (torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])):
                        raise AssertionError("render after step %d (%s) differs from a fresh network's: max |d rgb| %.3g" % (k, log[-1], float((got[0] - want[0]).abs().max())))
        except AssertionError as e:
            failures.append("case %d (seed %d) V=%d %dx%d after %d steps: %s; last steps: %s" % (c, s, V, H, W, len(log), e, log[-6:] if only is None else log))
        finally:
            uc_nerf_amd.set_inference_precision("bf16x3_fused")
            uc_nerf_amd.set_weight_cache("verify")
            uc_nerf_amd.set_split_operand("bf16")
            L.lib().ucnerf_set_fused_tail(1)
        if verbose:
            print("case %d (seed %d) V=%d %dx%d: %d steps %s" % (c, s, V, H, W, len(log), "FAIL" if failures and failures[-1].startswith("case %d " % c) else "ok"), flush=True)
    return dict(cases=cases, steps=steps, seed=seed, renders=renders, trains=trains, failures=failures, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--case", type=int, default=None, help="run this one case and print its whole step log on failure")
    ap.add_argument("--magnify", action="store_true", help="a render check after EVERY step (beware: the extra renders refresh caches and can hide a stale one)")
    a = ap.parse_args()
    s = run(a.cases, a.steps, a.seed, only=a.case, magnify=a.magnify)
    print("fuzz_dropin: %d cases x %d steps, %d renders and %d training steps checked against fresh networks, %d failures, %.0f s" % (
        s["cases"], s["steps"], s["renders"], s["trains"], len(s["failures"]), s["seconds"]))
    for f in s["failures"]:
        print("  " + f)
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
