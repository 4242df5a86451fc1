"""The oracle at the BENCHMARKED configuration (not gpu): oracle/model.py in fp32 against what the REAL reference produced
for MinkUNet / MinkUNetMs mk34 cr 1.0 on a 2 x 22k-voxel batch (tests/golden/model_mk34_*.npz `ref32_*`), and against its
own float64 evaluation stored next to it (`oracle64_*`)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import model as OM
from taseg_amd.data.synthetic import fill_parameters, make_model_cfg, strided_sample
from taseg_amd.pcseg.model import build_network


@pytest.mark.parametrize("name,in_dim,fname,training", [("MinkUNet", 4, "model_mk34_minkunet.npz", True),
                                                        ("MinkUNetMs", 5, "model_mk34_minkunet_ms.npz", False)])
def test_oracle_fp32_at_mk34_cr10(name, in_dim, fname, training):
    g = dict(np.load(os.path.join(GOLDEN, fname), allow_pickle=False))
    tag = "train_" if training else "eval_"
    torch.set_num_threads(4)
    cfg = make_model_cfg(name, in_dim=in_dim, cr=1.0)
    model = fill_parameters(build_network(cfg, 20), seed=3)
    learn = [n for n, _ in model.named_parameters()]
    params = {k: v.detach().clone().requires_grad_(k in learn) for k, v in model.state_dict().items()}
    om = OM.OracleMinkUNet(params, cfg, backend="numpy", training=training)
    fwd = om.forward_minkunet if name == "MinkUNet" else om.forward_minkunet_ms
    logits = fwd(g["coords"], torch.from_numpy(g["feats"]))
    loss = OM.loss_ce_lovasz(logits, torch.from_numpy(g["labels"]))
    loss.backward()
    got = logits.detach().numpy()[::8]
    assert np.abs(got - g["ref32_" + tag + "logits"]).max() <= 1e-3
    assert abs(float(loss) - float(g["ref32_" + tag + "loss"])) <= 1e-4
    names = g["param_names"].tolist()
    norms = np.array([float(params[n].grad.double().norm()) for n in names])
    n64, n32 = g["oracle64_" + tag + "gradnorms"], g["ref32_" + tag + "gradnorms"]
    ours, refs = np.abs(norms - n64) / np.maximum(n64, 1e-30), np.abs(n32 - n64) / np.maximum(n64, 1e-30)
    # fp32 noise level of this network: the reference's own worst distance to the float64 gradients (train-mode BatchNorm
    # over ~40 layers amplifies rounding differences of ANY fp32 evaluation: 1e-3 .. 8e-3 on the early layers)
    noise = max(np.linalg.norm(g[k] - g[k.replace("ref32_", "oracle64_")]) / np.linalg.norm(g[k.replace("ref32_", "oracle64_")])
                for k in g if k.startswith("ref32_" + tag + "grad/"))
    assert (ours <= max(2e-3, 2 * noise)).all()
    for k in [k for k in g if k.startswith("ref32_" + tag + "grad/")]:
        a = strided_sample(params[k.split("/", 1)[1]].grad.numpy(), 2048)
        want64, ref = g[k.replace("ref32_", "oracle64_")], g[k]
        e_ours = np.linalg.norm(a - want64) / np.linalg.norm(want64)
        e_ref = np.linalg.norm(ref - want64) / np.linalg.norm(want64)
        assert e_ours <= max(2e-3, 2 * noise), (k, e_ours, e_ref, noise)


def test_oracle_conv_on_dense_rulebook():
    """oracle rulebook + convolution against the reference on a cloud with 6.6 pairs per voxel (ops_dense.npz)"""
    from conftest import dense_ops_inputs
    from oracle import ts_oracle as O
    g = dict(np.load(os.path.join(GOLDEN, "ops_dense.npz"), allow_pickle=False))
    c = g["coords"]
    n = len(c)
    inp = dense_ops_inputs(n)
    _, nbmaps, nbsizes = O.build_kmap(c, c, O.get_kernel_offsets(3, 1, 1))
    assert np.array_equal(nbmaps, g["k3_nbmaps"]) and np.array_equal(nbsizes, g["k3_nbsizes"])
    for tag in ("full", "ragged"):
        x, w, gy = (inp[f"{tag}_{k}"].numpy() for k in ("x", "w", "gy"))
        y = O.conv_forward(x, w, nbmaps, nbsizes, (n, n))
        gx, gw = O.conv_backward(x, w, gy, nbmaps, nbsizes)
        for got, want in ((y[::4], g[f"{tag}_y"]), (gx[::4], g[f"{tag}_gx"]), (gw, g[f"{tag}_gw"])):
            assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    down = O.spdownsample(c, 2, 2, 1)
    assert np.array_equal(down, g["t_coords_d"])
    _, nb2, ns2 = O.build_kmap(c, down, O.get_kernel_offsets(2, 1, 1))
    x, wd, wu, gy = (inp[k].numpy() for k in ("t_x", "t_wd", "t_wu", "t_gy"))
    yd = O.conv_forward(x, wd, nb2, ns2, (n, len(down)))
    yu = O.conv_forward(yd, wu, nb2, ns2, (n, len(down)), transposed=True)
    assert np.abs(yd[::2] - g["t_yd"]).max() <= 1e-5 * np.abs(g["t_yd"]).max()
    assert np.abs(yu[::4] - g["t_yu"]).max() <= 1e-5 * np.abs(g["t_yu"]).max()
    gyd, gwu = O.conv_backward(yd, wu, gy, nb2, ns2, transposed=True)
    gx, gwd = O.conv_backward(x, wd, gyd, nb2, ns2)
    for got, want in ((gx[::4], g["t_gx"]), (gwd, g["t_gwd"]), (gwu, g["t_gwu"])):
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
