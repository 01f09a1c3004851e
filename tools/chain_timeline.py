"""In-kernel timeline of chain2 (diagnostic build path: SOW_AMD_CHAIN2_DEBUG bit 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
T, di, do = 32768, 512, 512
x = torch.randn(T, di, device="cuda", dtype=torch.bfloat16)
A = (torch.randn(di, 50, device="cuda") * 0.04).bfloat16(); B = (torch.randn(50, do, device="cuda") * 0.04).bfloat16()
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device="cuda")
for _ in range(3): ops.sow_forward(x, A, B, None, None, None, 1.0)
os.environ["SOW_AMD_CHAIN2_DBGBUF"] = hex(buf.data_ptr())
os.environ["SOW_AMD_CHAIN2_DEBUG"] = "32"
ops.sow_forward(x, A, B, None, None, None, 1.0)
torch.cuda.synchronize()
os.environ["SOW_AMD_CHAIN2_DEBUG"] = "0"
b = buf.cpu().view(256, 8, 16).double()
t0 = b[:, :, 0].min()
def rel(v): return ((v - t0) / 100.0)  # s_memtime ticks at 100 MHz -> us
for w in (0, 3, 4):
    s = b[:, w]
    print(f"wave {w}: start {rel(s[:,0]).median():.2f} | prologue done {rel(s[:,1]).median():.2f} | P1 end {rel(s[:,2]).median():.2f} | handoff end {rel(s[:,3]).median():.2f} | P2 end {rel(s[:,4]).median():.2f} | stores drained {rel(s[:,5]).median():.2f} | vmcnt-wait {s[:,8].median()/100:.2f} us | barrier-wait {s[:,9].median()/100:.2f} us")
print("block start spread (us):", float(rel(b[:, 0, 0]).max()), " last end:", float(rel(b[:, :4, 5]).max()))
