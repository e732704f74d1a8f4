"""Drop-in for reference checkerpose/model/init_lm.py (LM: one estimator shared by 13 objects).
Identical to init.py except that `p3d_normed` is (15, 3, N), the kNN table is (15, N, K) and forward takes the
1-based `obj_ids` (B,) that select each sample's graph (init_lm.py:64-66,110-119)."""
import torch

from .init import CONV1X1_IN_CHANS, InitNet_GNN as _InitNet_GNN, StaticGraph_module, knn  # noqa: F401


class InitNet_GNN(_InitNet_GNN):
    LM = True

    def forward(self, img, obj_ids, return_img_feats=False, return_graph_feats=False):
        res = self._run(img, obj_ids, want_feats=return_img_feats or return_graph_feats, want_graph=return_graph_feats)
        out = self._out_rows(res["bits"])          # res_log2 != 3 too (pretrain_lm.py:141 passes it through)
        if return_img_feats:
            return out, res["img_feats"]
        if return_graph_feats:
            return out, res["img_feats"], res["graph_feats"]
        return out
