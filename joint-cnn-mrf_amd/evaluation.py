"""Mirror of the reference's evaluation.py -- the detection-rate metric of MODEC (evaluation.py:4-37) -- and of the
batch loop that calls it, eval_error (main.py:275-283).

The arg-max over the heat maps is libjcm's kernel (first occurrence); the remaining arithmetic of det_rate is a
handful of [B,2,K] operations done with torch on the device."""
import torch


def det_rate_from_coords(pred, true, normalized_radius=10, joints='all'):
    """evaluation.py:26-37 on arg-max coordinates [B,2,K] (row, col)."""
    lhip_idx, rsho_idx = 0, 7                                        # evaluation.py:26
    pred, true = pred.to(torch.float32), true.to(torch.float32)
    torso = torch.linalg.norm(true[:, :, lhip_idx] - true[:, :, rsho_idx], dim=1, keepdim=True)     # [B,1]
    nd = torch.linalg.norm(pred - true, dim=1) * 100 / torso                                        # [B,K]
    if joints != 'all':
        nd = nd[:, list(joints)]
    return 100 * (nd <= normalized_radius).to(torch.float32).mean()


def det_rate(heat_map_pred, heat_map_target, normalized_radius=10, joints='all', engine=None):
    """evaluation.py:4-37.  heat_map_* [n_images, height, width, n_joints] torch CUDA fp32.
    Percentage of (image, joint) pairs whose predicted arg-max lies within `normalized_radius` % of the
    torso length (distance between channels 0 and 7 of the target, evaluation.py:26,29) of the target's."""
    if engine is None:
        from . import main as M
        engine = M.engine()
    pred = engine.argmax_coords(heat_map_pred)                       # [B,2,K]
    true = engine.argmax_coords(heat_map_target.contiguous())
    return float(det_rate_from_coords(pred, true, normalized_radius, joints))


def get_next_batch(X, Y, batch_size, shuffle=False, rng=None):
    """main.py:184-192: whole batches only -- the remainder len(X) % batch_size is dropped; `shuffle` draws a permutation."""
    import numpy as np
    n_batches = len(X) // batch_size
    idx = (rng or np.random).permutation(len(X))[:n_batches * batch_size] if shuffle else np.arange(len(X))[:n_batches * batch_size]
    for batch_idx in idx.reshape([n_batches, batch_size]):
        yield X[batch_idx], Y[batch_idx]


def eval_error(X_np, Y_np, engine, batch_size, use_sm=True, joints=(2,), det_radius=10):
    """main.py:275-283: run a data set through the tower in inference mode batch by batch and return the means over
    batches of (loss_pd, loss_sm, det_rate_pd, det_rate_sm).  X_np [N,480,720,3], Y_np [N,60,90,10] (numpy or torch,
    host or device); the remainder N % batch_size is dropped as in the reference (get_next_batch).  Everything stays on
    the device until the four means are read back."""
    n_batches = len(X_np) // batch_size
    if n_batches == 0:
        raise ValueError('eval_error needs at least one whole batch (%d examples, batch size %d)' % (len(X_np), batch_size))
    K = engine.n_joints
    acc = torch.zeros(4, dtype=torch.float64, device=engine.device)
    for bx, by in get_next_batch(X_np, Y_np, batch_size):
        x = torch.as_tensor(bx, dtype=torch.float32, device=engine.device).contiguous()
        y = torch.as_tensor(by, dtype=torch.float32, device=engine.device).contiguous()
        r = engine.eval_forward(x, y, use_sm=use_sm, want_prob=False)
        true = engine.argmax_coords(y[..., :K].contiguous())
        dr_pd = det_rate_from_coords(r['pd_coords'], true, det_radius, 'all' if joints == 'all' else list(joints))
        dr_sm = det_rate_from_coords(r['sm_coords'], true, det_radius, 'all' if joints == 'all' else list(joints)) if use_sm else dr_pd
        acc += torch.stack([r['losses'][0].double(), r['losses'][1].double(), dr_pd.double(), dr_sm.double()])
    return tuple(float(v) for v in (acc / n_batches).cpu())


def argmax_agreement(ref_prob, ref_coords, got_prob, got_coords, margin_mult=6.0, topk=32):
    """How far the arg-max coordinates of one arithmetic (`got_*`, e.g. a bf16 engine) are from another's (`ref_*`, the fp32 engine),
    evaluation.py:15-24 / main.py:389-397 being what the reference does with the heat maps.  prob [B,H,W,K] (softmax outputs), coords int32
    [B,2,K].  Returns, over all B*K joints and over the "safe" ones -- joints whose reference top-2 log-probability margin exceeds
    `margin_mult` x the measured rms log-probability error of `got` (taken on the reference's `topk` largest pixels per map, where
    the arg-max is decided) -- the exact-agreement rate, the rate within one heat-map cell (Chebyshev) and the mean Euclidean cell distance."""
    B, H, W, K = ref_prob.shape
    lr = torch.log(ref_prob.reshape(B, H * W, K).clamp_min(1e-37))
    lg = torch.log(got_prob.reshape(B, H * W, K).clamp_min(1e-37))
    top, idx = lr.topk(topk, dim=1)                                   # [B,topk,K]
    err = torch.gather(lg, 1, idx) - top
    rms = float(torch.sqrt((err.double() ** 2).mean()))
    margin = top[:, 0, :] - top[:, 1, :]                              # [B,K]
    safe = margin > margin_mult * rms
    d = (ref_coords.to(torch.int64) - got_coords.to(torch.int64))
    cheb = d.abs().amax(dim=1)                                        # [B,K]
    eucl = torch.sqrt((d.double() ** 2).sum(dim=1))

    def rates(mask):
        n = int(mask.sum())
        if n == 0:
            return {'n_joints': 0, 'exact': None, 'within1': None, 'mean_dist': None}
        return {'n_joints': n, 'exact': float((cheb[mask] == 0).double().mean()), 'within1': float((cheb[mask] <= 1).double().mean()),
                'mean_dist': float(eucl[mask].mean())}
    out = rates(torch.ones_like(safe))
    out.update({'rms_logprob_err': rms, 'max_logprob_err': float(err.abs().max()), 'margin_mult': margin_mult,
                'median_margin': float(margin.median()), 'safe': rates(safe)})
    return out
