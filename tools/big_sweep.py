import sys, json
sys.path.insert(0, "/root/repo")
import benchmarks_configs as bc, torch
torch.cuda.set_device(0)
for B, CH in ((128, 64), (128, 32), (256, 64), (256, 128), (128, 16)):
    r = bc.bench_big(B=B, CH=CH)
    print(B, CH, r["ms_per_step"], r["images_per_s"], r["e2e_frac_hbm"], flush=True)
    torch.cuda.empty_cache()
