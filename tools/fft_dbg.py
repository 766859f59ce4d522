import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joint_cnn_mrf_amd.engine import Engine
from oracle import jcm_oracle as O
rs = np.random.RandomState(1)
B, H, W, cin, cout = 1, 30, 45, 128, 128
p = {'c/weights': (rs.standard_normal((9, 9, cin, cout)) * np.sqrt(2.0 / (81 * cin))).astype(np.float32), 'c/biases': (0.1 * rs.standard_normal(cout)).astype(np.float32)}
x = rs.standard_normal((B, H, W, cin)).astype(np.float32)
eng = Engine(device=0).load_params(p)
print('kernel', eng.conv_kernel_name('c', B, H, W), flush=True)
got = eng.conv_layer(torch.as_tensor(x, device='cuda:0'), 'c', 1, last_layer=True, n_out=cout).cpu().numpy()
ref = O.conv_layer(x.astype(np.float64), p, 9, 1, 'c', last_layer=True)
print('err', np.abs(got - ref).max() / np.abs(ref).max())
