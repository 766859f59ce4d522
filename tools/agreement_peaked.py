"""Arg-max agreement of the bf16 engines with the fp32 engine on PEAKED heat maps: the full-width network is trained with this
repo's own trainer on a small synthetic set (U[0,1) images, 3x3 binomial target blobs, data.py:112-114) until its heat maps are shaped like a trained
reference's -- one dominant peak per joint --, then the trained parameters run through the fp32 and the bf16 engines on the training images.
    python tools/agreement_peaked.py [steps=300] [images=32]"""
import json, sys
import numpy as np, torch
sys.path.insert(0, '.')
import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from joint_cnn_mrf_amd.engine import Engine
from joint_cnn_mrf_amd.evaluation import argmax_agreement
from joint_cnn_mrf_amd.train import Trainer


def train_peaked(steps=300, n_images=32, batch=8, seed=0, verbose=True):
    p = synth.make_pd_params(debug=False)      # full width: the bf16 kernels take channel counts in multiples of 32, which --debug's 16 are not
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
    data = [(torch.as_tensor(synth.make_images(batch, seed=300 + seed + i), device='cuda:0'), torch.as_tensor(synth.make_targets(batch, seed=400 + seed + i), device='cuda:0'))
            for i in range(n_images // batch)]
    eng = Engine(device=0).load_params(p)
    tr = Trainer(eng, optimizer='adam', lr=0.001, lmbd=0.0, use_sm=True, n_updates_total=steps)
    for s in range(steps):
        x, y = data[s % len(data)]
        losses, _ = tr.train_step(x, y)
        if verbose and (s % 50 == 0 or s == steps - 1):
            l = losses.cpu().numpy()
            print('step %d  pd loss %.4f  sm loss %.4f' % (s, float(l[1]), float(l[2])), flush=True)
    trained = tr.get_params(p)
    eng.close()
    return trained, torch.cat([d[0] for d in data]), torch.cat([d[1] for d in data])


def measure(trained, x, y):
    torso = y[..., 9:].contiguous()
    eng = Engine(device=0).load_params(trained)
    ref = eng.forward(x, torso, use_sm=True)
    ref = {k: v.clone() for k, v in ref.items()}
    eng.close()
    # how well the fp32 engine found the targets, and how peaked its maps are
    tgt = y[..., :9].reshape(y.shape[0], -1, 9).argmax(dim=1)
    tcoords = torch.stack([tgt // 90, tgt % 90], dim=1).to(torch.int32)
    out = {'fp32_hits_target': float((ref['pd_coords'] == tcoords).all(dim=1).double().mean()),
           'pd_peak_prob_median': float(ref['pd_prob'].reshape(x.shape[0], -1, 9).amax(dim=1).median())}
    for name, kw in {'default': {}, 'strict': dict(fft_single=False, fft_t16=False), 'direct': dict(conv9_fft=False)}.items():
        e = Engine(device=0, precision='bf16', **kw).load_params(trained)
        got = e.forward(x, torso, use_sm=True)
        out[name] = {'pd': argmax_agreement(ref['pd_prob'], ref['pd_coords'], got['pd_prob'], got['pd_coords']),
                     'sm': argmax_agreement(ref['sm_prob'], ref['sm_coords'], got['sm_prob'], got['sm_coords'])}
        e.close()
    return out


if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    trained, x, y = train_peaked(steps, n)
    print(json.dumps(measure(trained, x, y), indent=1))
