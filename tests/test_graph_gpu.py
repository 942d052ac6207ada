"""HIP-graph replay of model.inference (star_amd/graph.py): bit-identical to the eager call on the same inputs and mask
noise, follows new inputs through its static buffers, re-captures when shapes or weights change, draws fresh mask noise
on every replay."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from star_amd.graph import GraphedInference
    from star_amd.models import VQSTAR, multiagent_mae
    from star_amd.synthetic import make_batch
    from star_amd.utils.CoModule import CoModule

DEV = "cuda"


def _args(data):
    return CoModule._unpack(data)


@pytest.mark.parametrize("T,mask", [(1, "complement"), (2, "random")])
def test_graph_replay_matches_eager(T, mask):
    torch.manual_seed(0)
    model = multiagent_mae.amortized_ind_patch8_light(time_stamp=T, mask_method=mask).to(DEV).eval()
    B = 2
    L = (256 // 8) ** 2
    model._noise_override = torch.rand(5 * B * (T if mask == "random" else 1), L, device=DEV)     # same masks in both runs
    run = GraphedInference(model, B, mask_ratio=0.5)
    for seed in (1, 2, 3):
        data = make_batch(B, 5, T, seed=seed, device=DEV)
        bev, nxt, teacher, trans, na = _args(data)
        with torch.no_grad():
            want = model.inference(bev, nxt, teacher, trans, na, B, mask_ratio=0.5)
        got = run(bev, nxt, teacher, trans, na)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert torch.equal(g, w)
    assert run.captures == 1
    # another batch size -> new signature -> one more capture; weights changed -> one more
    data = make_batch(1, 5, T, seed=9, device=DEV)
    model._noise_override = model._noise_override[:5 * (T if mask == "random" else 1)].clone()
    run1 = GraphedInference(model, 1, mask_ratio=0.5)
    bev, nxt, teacher, trans, na = _args(data)
    a = [t.clone() for t in run1(bev, nxt, teacher, trans, na)]
    with torch.no_grad():
        model.decoder_pred_occ.bias.add_(0.5)
    b = run1(bev, nxt, teacher, trans, na)
    assert run1.captures == 2 and not torch.equal(a[0], b[0])
    with torch.no_grad():
        want = model.inference(bev, nxt, teacher, trans, na, 1, mask_ratio=0.5)
    assert torch.equal(b[0], want[0]) and torch.equal(b[1], want[1])


def test_graph_replay_draws_fresh_noise_and_comodule_path():
    torch.manual_seed(0)
    model = multiagent_mae.amortized_ind_patch8_light(time_stamp=2, mask_method="random").to(DEV).eval()
    cm = CoModule(model, None, "ind_mae", graph_inference=True)
    data = make_batch(1, 5, 2, seed=4, device=DEV)
    losses = [cm.infer_mae_completion(data, 1, 0.75)[0] for _ in range(4)]
    assert len(set(losses)) > 1                        # random masks differ between replays
    assert all(abs(l - losses[0]) < 0.05 for l in losses)
    loss, result, ind, latent = cm.infer_mae_completion(data, 1, 0.75)
    assert tuple(result.shape) == (5, 13, 256, 256) and set(torch.unique(result).tolist()) <= {0.0, 1.0}
    # ego slot of the fusion = the agent's own input wherever the input is occupied (ego_late_fusion)
    bev = CoModule._unpack(data)[0]
    assert bool((result >= bev).all())


def test_graph_replay_vqstar():
    torch.manual_seed(0)
    model = VQSTAR.vqstar(time_stamp=1, mask_method="complement", num_vq_embeddings=256, vq_embedding_dim=32).to(DEV).eval()
    model._noise_override = torch.rand(5, 1024, device=DEV)
    data = make_batch(1, 5, 1, seed=6, device=DEV)
    bev, nxt, teacher, trans, na = _args(data)
    run = GraphedInference(model, 1)
    with torch.no_grad():
        want = model.inference(bev, nxt, teacher, trans, na, 1)
    got = run(bev, nxt, teacher, trans, na)
    for g, w in zip(got, want):
        assert torch.equal(g, w)
