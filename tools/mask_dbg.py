import sys, torch
sys.path.insert(0, "/root/repo")
from bs_yolo_amd import masks as HM
from oracle import postproc_ref as PP
DEV = "cuda:0"
g = torch.Generator().manual_seed(8)
nm, mh, mw, ih, iw, n = 32, 40, 48, 160, 192, 13
protos = torch.randn(nm, mh, mw, generator=g)
coef = torch.randn(n, nm, generator=g)
xy = torch.rand(n, 2, generator=g) * torch.tensor([iw * 0.6, ih * 0.6])
wh = torch.rand(n, 2, generator=g) * torch.tensor([iw * 0.4, ih * 0.4]) + 4
boxes = torch.cat((xy, xy + wh), 1)
ref = PP.process_mask(protos.float(), coef, boxes.clone(), (ih, iw), False)
got = HM.process_mask(protos.to(DEV), coef.to(DEV), boxes.to(DEV), (ih, iw), False).cpu()
d = (got != ref)
print("mismatch", d.float().mean().item(), "per mask", d.flatten(1).sum(1).tolist())
print("ref ones per mask", ref.flatten(1).sum(1).tolist())
print("got ones per mask", got.flatten(1).sum(1).tolist())
m = int(d.flatten(1).sum(1).argmax())
nz = d[m].nonzero()
print("mask", m, "box*ratio", (boxes[m] * torch.tensor([mw / iw, mh / ih, mw / iw, mh / ih])).tolist(), "mismatch rows", sorted(set(nz[:, 0].tolist())), "cols", sorted(set(nz[:, 1].tolist()))[:50])
