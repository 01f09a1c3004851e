"""In-kernel timeline of chain2 (diagnostic build path: SOW_AMD_CHAIN2_DEBUG bit 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
T, di, do = 32768, 512, 512
x = torch.randn(T, di, device="cuda", dtype=torch.bfloat16)
A = (torch.randn(di, 50, device="cuda") * 0.04).bfloat16(); B = (torch.randn(50, do, device="cuda") * 0.04).bfloat16()
buf = torch.zeros(512 * 4 * 16, dtype=torch.int64, device="cuda")
for _ in range(3): ops.sow_forward(x, A, B, None, None, None, 1.0)
os.environ["SOW_AMD_CHAIN2_DBGBUF"] = hex(buf.data_ptr())
os.environ["SOW_AMD_CHAIN2_DEBUG"] = "32"
ops.sow_forward(x, A, B, None, None, None, 1.0)
torch.cuda.synchronize()
os.environ["SOW_AMD_CHAIN2_DEBUG"] = "0"
b = buf.cpu().view(512, 4, 16).double()
t0 = b[:, :, 0].min()
def rel(v): return ((v - t0) / 100.0)  # s_memtime ticks at 100 MHz -> us
for w in (0, 1, 2):
    s = b[:, w]
    print(f"wave {w}: start {rel(s[:,0]).median():.2f} | prologue done {rel(s[:,1]).median():.2f} | P1 end {rel(s[:,2]).median():.2f} | handoff end {rel(s[:,3]).median():.2f} | P2 end {rel(s[:,4]).median():.2f} | stores drained {rel(s[:,5]).median():.2f} | vmcnt-wait {s[:,8].median()/100:.2f} us | barrier-wait {s[:,9].median()/100:.2f} us")
first, second = b[:256], b[256:]
print("second-half workgroups start later by (ticks, median):", float((second[:, 0, 0].median() - first[:, 0, 0].median())))
print("kernel span (ticks):", float(b[:, :2, 5].max() - b[:, :, 0].min()))

print("per-half phase durations in ticks (median): prologue / P1 / handoff / P2 / drain")
bb = buf.cpu().view(512, 4, 16).double()
for name, sl_ in (("first half (b < 256)", slice(0, 256)), ("second half", slice(256, 512))):
    w0 = bb[sl_, 0]
    d = [(w0[:, i + 1] - w0[:, i]).median().item() for i in range(5)]
    print(name, [int(v) for v in d], "start rel. to kernel t0:", int((w0[:, 0] - bb[:, :, 0].min()).median()), " end:", int((w0[:, 5] - bb[:, :, 0].min()).median()))
import collections
hw = buf.cpu().view(512, 4, 16)[:, 0, 10].tolist(); xcc = buf.cpu().view(512, 4, 16)[:, 0, 11].tolist()
loc = collections.defaultdict(list)
for bi, (h, x) in enumerate(zip(hw, xcc)):
    cu = (h >> 8) & 0xf; sh = (h >> 12) & 1; se = (h >> 13) & 7
    loc[(x & 0xf, se, sh, cu)].append(bi)
print("distinct (xcc,se,sh,cu):", len(loc))
for k in list(loc)[:12]: print(k, loc[k])
pairs = collections.Counter()
for k, v in loc.items():
    if len(v) == 2: pairs[v[1] - v[0]] += 1
print("block-id distance of co-resident pairs:", pairs.most_common(8))
