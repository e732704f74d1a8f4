"""micro-benchmark of cp_hr_stem (NCHW fp32 crop -> conv1/s2 -> conv2/s2, 256 x 256 -> 64 x 64 x 64)"""
import sys
import torch
sys.path.insert(0, ".")
from checkerpose_amd import _abi
lib = _abi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img = torch.randn(B, 3, 256, 256, device=dev)
w1 = (torch.randn(64, 3, 3, 3, device=dev) * 0.1).contiguous(); w2 = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous()
p1 = torch.empty(lib.cp_hr_stem_weight_bytes(0), dtype=torch.uint8, device=dev)
p2 = torch.empty(lib.cp_hr_stem_weight_bytes(1), dtype=torch.uint8, device=dev)
_abi.check(lib.cp_pack_hr_stem_weights(st, w1.data_ptr(), w2.data_ptr(), p1.data_ptr(), p2.data_ptr()))
a = [torch.ones(64, device=dev) for _ in range(4)]
out = torch.empty(B, 64, 64, 64, device=dev, dtype=torch.bfloat16)
run = lambda: _abi.check(lib.cp_hr_stem(st, img.data_ptr(), B, 256, 256, p1.data_ptr(), a[0].data_ptr(), a[1].data_ptr(), p2.data_ptr(),
                                        a[2].data_ptr(), a[3].data_ptr(), out.data_ptr()))
for rep in range(3):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("hr_stem B=%d: %7.1f us  %5.2f TB/s" % (B, us, (img.numel() * 4 + out.numel() * 2) / us / 1e6), flush=True)
