"""The generation loop of bench.py on its own (secondary metric "gen lines/sec"): HWWithStyle.forward via generate_stream, B lines per call.
  python tools/gen_bench.py [B] [calls]      (under rocprofv3 --kernel-trace --stats for the per-kernel view)"""
import sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import ops, rng
from handwriting_line_generation_amd.generate import generate_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 48
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
model = tr.model; model.eval()
g = torch.Generator().manual_seed(4321)
labels = [torch.randint(1, cfg["model"]["num_class"], (30, B), generator=g, dtype=torch.int32) for _ in range(8)]
lengths = torch.IntTensor([30] * B)
styles = [ops.h2d(torch.randn(B, cfg["model"]["style_dim"], generator=g), tr.gpu) for _ in range(8)]
reqs = lambda n: ((labels[i % 8], lengths, styles[i % 8]) for i in range(n))   # noqa: E731
with torch.no_grad():
    for img, _ in generate_stream(model, reqs(8)):
        pass
    torch.cuda.synchronize()
    t = time.perf_counter(); px = 0
    for img, _ in generate_stream(model, reqs(calls)):
        px += img.shape[3]
    torch.cuda.synchronize()
    t = time.perf_counter() - t
print("B %d: %.0f lines/s, %.3f ms per call, mean width %.0f px" % (B, B * calls / t, t / calls * 1e3, px / calls))
