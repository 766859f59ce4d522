"""joint-cnn-mrf_amd: MI355X-native joint-heat-map inference (part detector + MRF spatial
model) behind the call surface of max-andr/joint-cnn-mrf's main.py.

Only the hot path lives here: `csrc/` (HIP kernels + the C-ABI library `libjcm.so`),
`_lib` (ctypes binding), `engine` (handle object), `main` (the reference-shaped host
module: model / conv_mrf / spatial_model / spatial_softmax), `dist` (batch sharding +
coords all-gather), `synth` / `priors` (synthetic parameters and the prior format)."""
