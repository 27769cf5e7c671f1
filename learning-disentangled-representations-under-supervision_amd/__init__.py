"""MI355X-native CGN (Hu et al. 2017) wake-sleep training step.

Layout mirrors the reference repo so that it drops in for the CGN forward/backward path:
  model/{encoder,generator,decoder,discriminator_sentiment,cgn}.py, selfModules/{embedding,tdnn,highway}.py,
  utils/{config,functional}.py, train_functions.py
plus csrc/ (hand-written gfx950 HIP kernels + the C-ABI of include/cgn_hip.h) and hip/ (ctypes binding).
"""
