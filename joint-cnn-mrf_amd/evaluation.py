"""Mirror of the reference's evaluation.py: the detection-rate metric of MODEC (evaluation.py:4-37).

The argmax over the heat maps is libjcm's kernel (`jcm_argmax_coords`, first occurrence); the
remaining arithmetic is a handful of [B,2,K] operations done with torch on the device."""
import torch


def det_rate(heat_map_pred, heat_map_target, normalized_radius=10, joints='all', engine=None):
    """evaluation.py:4-37.  heat_map_* [n_images, height, width, n_joints] torch CUDA fp32.
    Percentage of (image, joint) pairs whose predicted arg-max lies within `normalized_radius` % of the
    torso length (distance between channels 0 and 7 of the target, evaluation.py:26,29) of the target's."""
    if engine is None:
        from . import main as M
        engine = M.engine()
    lhip_idx, rsho_idx = 0, 7                                        # evaluation.py:26
    pred = engine.argmax_coords(heat_map_pred).to(torch.float32)     # [B,2,K]
    true = engine.argmax_coords(heat_map_target.contiguous()).to(torch.float32)
    torso = torch.linalg.norm(true[:, :, lhip_idx] - true[:, :, rsho_idx], dim=1, keepdim=True)     # [B,1]
    nd = torch.linalg.norm(pred - true, dim=1) * 100 / torso                                        # [B,K]
    if joints != 'all':
        nd = nd[:, list(joints)]
    return float(100 * (nd <= normalized_radius).to(torch.float32).mean())
