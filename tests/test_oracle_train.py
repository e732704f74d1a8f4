"""CPU: the training-side oracle (oracle/train_oracle.py) against the vectors the REFERENCE's modules produced
(tests/golden/train_ops.npz, generator tests/golden/make_golden_train.py)."""
import os

import numpy as np
import pytest
import torch

from tests import train_cases as TC
from oracle import train_oracle as TO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_ops.npz"))


@pytest.mark.parametrize("name", list(TC.CODE_CASES))
def test_code_loss_matches_reference(name):
    c = TC.CODE_CASES[name]
    pred, gt, mask = TC.code_inputs(c)
    loss, grad = TO.code_loss(pred.numpy(), gt.numpy(), None if mask is None else mask.numpy(), c["type"])
    np.testing.assert_allclose(loss, G["code_%s_loss" % name], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(grad, G["code_%s_grad" % name], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", list(TC.MASK_CASES))
def test_mask_loss_matches_reference(name):
    c = TC.MASK_CASES[name]
    pred, gt = TC.mask_inputs(c)
    loss, grad = TO.mask_loss_interpolate(pred[:, c["ch"]:c["ch"] + 1].numpy(), gt.numpy())
    np.testing.assert_allclose(loss, G["mask_%s_loss" % name], rtol=2e-6)
    full = np.zeros(pred.shape)
    full[:, c["ch"]] = grad[:, 0]
    np.testing.assert_allclose(full, G["mask_%s_grad" % name], rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("name", list(TC.EDGE_CASES))
def test_edgeconv_backward_matches_reference(name):
    c = TC.EDGE_CASES[name]
    x, idx, gup = TC.edge_inputs(c)
    wpq, sc, sh = TC.edge_folded_weights(c)                     # (2C', Cin), (2C',), (2C',)
    pq = (torch.einsum("bcn,oc->bno", x.double(), wpq.double()) * sc.double() + sh.double()).numpy()
    out, _ = TO.edgeconv_gather_max(pq, idx.numpy(), c["slope"])
    np.testing.assert_allclose(out.transpose(0, 2, 1), G["edge_%s_out" % name], rtol=1e-5, atol=2e-6)
    dpq = TO.edgeconv_gather_max_bwd(pq, idx.numpy(), gup.numpy().transpose(0, 2, 1), c["slope"])
    dx = np.einsum("bno,oc->bcn", dpq * sc.double().numpy(), wpq.double().numpy())
    np.testing.assert_allclose(dx, G["edge_%s_dx" % name], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("name", list(TC.I2F_CASES))
def test_index2feat_backward_matches_reference(name):
    c = TC.I2F_CASES[name]
    patches, x_id, y_id, mask, gup = TC.i2f_inputs(c)
    B, E, Hp, Wp = patches.shape
    d = TO.index2feat_gather_bwd(gup.numpy(), x_id.numpy(), y_id.numpy(), mask.numpy(), Hp, Wp, E, c["k"])
    np.testing.assert_allclose(d.transpose(0, 3, 1, 2), G["i2f_%s_dpatches" % name], rtol=1e-5, atol=1e-7)


def test_nearest_index_is_torch_nearest():
    for n_in, n_out in ((128, 64), (64, 64), (100, 64), (90, 64), (37, 64), (480, 64)):
        ref = torch.nn.functional.interpolate(torch.arange(n_in, dtype=torch.float32)[None, None, None], size=(1, n_out),
                                              mode="nearest")[0, 0, 0].long().numpy()
        np.testing.assert_array_equal(TO.nearest_index(n_out, n_in), ref)
