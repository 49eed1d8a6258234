import sys, torch
sys.path.insert(0, "/root/repo")
from bs_yolo_amd import ops as O
DEV = "cuda:0"
def h16(t): return t.half().float()
B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 16, 32)
cin, c, c2 = 64, 32, 128
g = torch.Generator().manual_seed(33)
def wt(co, ci, k):
    return h16(torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5), torch.randn(co, generator=g) * 0.2
(w1, b1), (wa, ba), (wb, bb), (w4, b4) = wt(2 * c, cin, 1), wt(c // 2, c, 3), wt(c, c // 2, 3), wt(c2, 3 * c, 1)
buf = h16(torch.randn(B, H, W, cin + 16, generator=g)).half().to(DEV)
x = buf[..., 8:8 + cin]
out = O.c3k2_fused(x, w1, b1, wa, ba, wb, bb, w4, b4)
p1, pb1 = O.pack_conv_weight(w1, b1, DEV)
p4, pb4 = O.pack_conv_weight(w4, b4, DEV)
cat = torch.zeros(B, H, W, 3 * c, dtype=torch.float16, device=DEV)
cat[..., :2 * c] = O.conv2d_nhwc(x.contiguous(), p1, pb1, 2 * c, 1, 1, True)
O.bottleneck_fused(cat[..., c:2 * c], wa, ba, wb, bb, out=cat[..., 2 * c:])
three = O.conv2d_nhwc(cat, p4, pb4, c2, 1, 1, True)
torch.cuda.synchronize()
d = (out.float() - three.float()).abs()
nz = (d > 0).nonzero()
print("differing", len(nz), "of", d.numel(), "max", float(d.max()), "max|ref|", float(three.float().abs().max()))
print("first", nz[:10].tolist())
import collections
print("by channel%32", sorted(collections.Counter((nz[:, 3] % 32).tolist()).items())[:40])
print("by y", sorted(collections.Counter(nz[:, 1].tolist()).items()))
print("by x", sorted(collections.Counter(nz[:, 2].tolist()).items()))
# which K segment: zero out segments of w4
for seg in range(3):
    w4z = w4.clone(); 
    for s2 in range(3):
        if s2 != seg: w4z[:, 32*s2:32*s2+32] = 0
    o2 = O.c3k2_fused(x, w1, b1, wa, ba, wb, bb, w4z, b4)
    pz, pbz = O.pack_conv_weight(w4z, b4, DEV)
    t2 = O.conv2d_nhwc(cat, pz, pbz, c2, 1, 1, True)
    dd = (o2.float() - t2.float()).abs()
    print("segment", seg, "only: differing", int((dd > 0).sum()), "max", float(dd.max()))
