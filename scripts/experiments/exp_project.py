import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from computervisionimagestich2_amd import capi
dev = torch.device("cuda:0")
out = []
for (w, h, td) in [(4096, 4096, torch.uint8), (4096, 4096, torch.float32), (384, 512, torch.uint8)] + ([(4096, 3072, torch.uint8), (4096, 3072, torch.float32)] if os.environ.get("EXP_LANDSCAPE") else []):
    src = capi.dev_synth(w, h, 0, td, dev); dst = torch.empty_like(src)
    for _ in range(3): capi.dev_project(src, 15.0, dst)
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(30): capi.dev_project(src, 15.0, dst)
    b.record(); torch.cuda.synchronize(); out.append(f"{w}x{h} {'u8' if td == torch.uint8 else 'f32'} {a.elapsed_time(b) / 30:.4f}")
print(os.environ.get("STITCH_LIB", "default").split("libstitch_")[-1], " | ".join(out))
