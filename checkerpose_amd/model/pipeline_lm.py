"""Drop-in for reference checkerpose/model/pipeline_lm.py:342-425 (LM shared estimator): PoseNet_GNNskip whose
forward takes per-sample 1-based `obj_ids`; each sample gathers along its own object's kNN graph
(`self.knn_idx[obj_ids-1]`, pipeline_lm.py:55-57).  The ablation classes (*_ABwoProg) are out of scope
(SURVEY.md §2 row 4)."""
from .pipeline import (IMG_FEATS_DIMS, Index2Feat_module, MLP_QueryNet, PoseNet_GNNskip as _PoseNet_GNNskip,  # noqa: F401
                       Refine_moduleGNN, StaticGraph_module, get_gdrn_upsample_module, get_MLP_leakyReLU_layers, knn)


class PoseNet_GNNskip(_PoseNet_GNNskip):
    LM = True

    def forward(self, img, p3d_normed, obj_ids, stage=None):
        active = stage if stage is not None else self.num_refine_steps
        res = self._run(img, obj_ids, stage=stage)
        return self._outputs(res, active)
