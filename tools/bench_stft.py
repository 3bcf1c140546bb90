"""STFT / mel front-end micro-benchmark: y ~ U(-0.5,0.5)[16, 262144] -> 16 x 1025 frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd

dev = "cuda"
st = ctts_amd.TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000).to(dev)
g = torch.Generator().manual_seed(1)
y = (torch.rand(16, 262144, generator=g) - 0.5).to(dev)
for _ in range(3):
    mel, en = st.mel_spectrogram(y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    mel, en = st.mel_spectrogram(y)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
frames = mel.shape[0] * mel.shape[2]
print(f"mel_spectrogram: {frames} frames in {us:.1f} us = {frames / us:.2f} M frames/s; DFT-as-GEMM {2.18e6 * frames / us / 1e6:.1f} TFLOP/s of 157.3 "
      f"(fp32 MFMA); algorithmic bytes {frames * 1348 / us / 1e3:.1f} GB/s")
