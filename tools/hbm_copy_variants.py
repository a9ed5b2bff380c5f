"""The HBM copy yardstick of bench.py (yardstick_hbm_copy) at three sizes next to torch's own copy_; ANCSH_COPY_VARIANT=0|1|2 selects the kernel
(tools/microbench/membw.hip, its own libyardstick.so).  tools/capture_profiles.sh copy -> profiles/*_hbm_copy_variants.txt."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
for mib in (256, 1024, 4096):
    print(os.environ.get("ANCSH_COPY_VARIANT", "0"), mib, bench.measured_hbm_copy(dev, mib=mib), flush=True)
a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
print("torch copy_", round(2.0 * (1 << 30) * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1))
