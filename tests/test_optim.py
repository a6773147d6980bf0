"""The library's Adam against torch.optim.Adam - the optimiser the reference constructs at
main.py:262 - run on CPU tensors as the checker."""
import numpy as np
import pytest
import torch

import semantichuman_amd as sh

SHAPES = [(7,), (33, 5), (4096,), (4097,), (3, 16, 30), (128, 1024), (1,)] + [(11 + i,) for i in range(60)]   # > 60 tensors: two launches


def make(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in SHAPES]


def set_grads(params, k):
    g = torch.Generator().manual_seed(100 + k)
    for p in params:
        p.grad = (torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))).to(p.device)


def test_constructor_validation():
    w = [torch.zeros(3, requires_grad=True)]
    with pytest.raises(NotImplementedError):
        sh.optim.Adam(w, amsgrad=True)
    with pytest.raises(ValueError):
        sh.optim.Adam(w, betas=(0.9, 1.0))
    opt = sh.optim.Adam(w, lr=1e-3, weight_decay=5e-5)
    assert opt.param_groups[0]["lr"] == 1e-3 and opt.param_groups[0]["weight_decay"] == 5e-5
    w[0].grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        opt.step()


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 5e-5])
def test_hip_adam_matches_torch_adam(wd):
    dev = torch.device("cuda:0")
    mine, ref = make(dev), make("cpu")
    o1 = sh.optim.Adam(mine, lr=1e-3, weight_decay=wd)
    o2 = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd)
    s1 = torch.optim.lr_scheduler.StepLR(o1, 2, gamma=0.5)
    s2 = torch.optim.lr_scheduler.StepLR(o2, 2, gamma=0.5)
    for k in range(6):
        set_grads(mine, k); set_grads(ref, k)
        o1.step(); o2.step(); s1.step(); s2.step()
        for a, b in zip(mine, ref):
            # a step moves a weight by <= lr; agreement to ~1 ulp of the weights (<= 1e-6 abs per step)
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), atol=2e-7 * (k + 1), rtol=2e-7 * (k + 1))
    assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"]
    for a, b in zip(mine, ref):
        assert float(o1.state[a]["step"]) == float(o2.state[b]["step"]) == 6
        # exp_avg = lerp(exp_avg, g, 0.1) of O(1) gradients: agreement to an ulp of the terms, not of a cancelled result
        np.testing.assert_allclose(o1.state[a]["exp_avg"].cpu().numpy(), o2.state[b]["exp_avg"].numpy(), atol=1e-7, rtol=2e-6)
        np.testing.assert_allclose(o1.state[a]["exp_avg_sq"].cpu().numpy(), o2.state[b]["exp_avg_sq"].numpy(), atol=1e-12, rtol=2e-6)


@pytest.mark.gpu
def test_hip_adam_state_dict_interchange_with_torch():
    dev = torch.device("cuda:0")
    a, b = make(dev), make(dev)
    mine, ref = sh.optim.Adam(a, lr=1e-3, weight_decay=5e-5), torch.optim.Adam(b, lr=1e-3, weight_decay=5e-5)
    for k in range(2):
        set_grads(a, k); set_grads(b, k)
        mine.step(); ref.step()
    sd_m, sd_r = mine.state_dict(), ref.state_dict()
    assert sorted(sd_m.keys()) == sorted(sd_r.keys()) and sorted(sd_m["state"][0].keys()) == sorted(sd_r["state"][0].keys())
    # resume each from the OTHER's checkpoint: one more step must land on the same weights
    mine2, ref2 = sh.optim.Adam(a, lr=1.0), torch.optim.Adam(b, lr=1.0)
    mine2.load_state_dict(sd_r); ref2.load_state_dict(sd_m)
    assert mine2.param_groups[0]["lr"] == 1e-3
    set_grads(a, 9); set_grads(b, 9)
    mine2.step(); ref2.step()
    for p, q in zip(a, b):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), atol=1e-6, rtol=1e-6)
        assert float(mine2.state[p]["step"]) == 3


@pytest.mark.gpu
def test_hip_adam_overlapped_update_is_identical():
    """Updating big parameters from the post-accumulate-grad hook (side stream, under the rest of backward) gives
    bit-identical weights to updating everything in step()."""
    dev = torch.device("cuda:0")
    outs = []
    for overlap in (False, True):
        params = make(dev)
        opt = sh.optim.Adam(params, lr=1e-3, weight_decay=5e-5)
        if overlap:
            opt.overlap_backward(min_numel=4096)
            assert len(opt._hooks) == 3
        for k in range(3):
            opt.zero_grad(set_to_none=True)
            x = torch.full((), 0.5 + k, device=dev)
            loss = sum((p * p).sum() * x + p.sum() for p in params)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for p in params])
    for p, q in zip(*outs):
        assert torch.equal(p, q)


@pytest.mark.gpu
def test_hip_adam_captured_step_follows_steplr_through_sync_lr():
    """a step captured in a hipGraph reads the learning rate from a device scalar; a replay never re-enters
    step(), so the scheduler's change reaches it through sync_lr().  Replayed-with-sync == eager, bit for bit."""
    dev = torch.device("cuda:0")
    outs = []
    for captured in (False, True):
        params = make(dev)
        set_grads(params, 0)
        opt = sh.optim.Adam(params, lr=1e-3, weight_decay=5e-5)
        sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)
        graph = None
        if captured:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                opt.step()                                    # warm-up: creates the state
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                opt.step()
            # undo the two updates the warm-up and the capture... capture does not execute; undo the warm-up step
            fresh = make(dev)
            for p, q in zip(params, fresh):
                p.data.copy_(q)
            for st in opt.state.values():
                for v in st.values():
                    v.zero_()
        for k in range(4):
            if graph is not None:
                graph.replay()
            else:
                opt.step()
            sched.step()
            opt.sync_lr()
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for p in params])
        assert opt.param_groups[0]["lr"] == pytest.approx(1e-3 * 0.5 ** 4)
    for p, q in zip(*outs):
        assert torch.equal(p, q)
