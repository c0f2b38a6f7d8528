"""Drop-in `indextts` package backed by the MI355X HIP engine (libitts_hip).

Same import paths and call signatures as the reference's hot-path modules
(/root/reference/indextts/{infer,cli}.py, gpt/model.py, BigVGAN/models.py,
BigVGAN/alias_free_activation/cuda/{load,activation1d}.py, vqvae/xtts_dvae.py); the arithmetic runs in
hand-written HIP kernels through the C ABI of include/itts_hip.h.  Put `index-tts-ipex_amd/` on sys.path
in place of the reference tree."""
