"""Body of __graft_entry__.smoke(): one small pass through HIP kernels of the hot path on cuda:0, checked
against the CPU oracle (conv -> RoIAlign -> NMS)."""
import torch
import torch.nn.functional as F


def run(pkg, dev):
    from oracle import c_binding as CO
    ops = pkg.ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 64, 32, 48, generator=g).half()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half()
    b = torch.randn(64, generator=g)
    y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b.to(dev), 1, 1, relu=True)
    ref = F.relu(F.conv2d(x.float(), w.float(), b, padding=1)).half()
    err = (y.cpu().permute(0, 3, 1, 2).float() - ref.float()).abs().max().item()
    assert err < 2e-2, f"conv mismatch {err}"
    boxes = torch.tensor([[8.0, 8.0, 100.0, 90.0], [30.0, 20.0, 180.0, 120.0], [33.0, 22.0, 182.0, 118.0]])
    bidx = torch.zeros(3, dtype=torch.int32)
    pooled = ops.roi_align([y], (0.25,), boxes.to(dev), bidx.to(dev), 7, torch.float32, min_level=2).cpu().permute(0, 3, 1, 2)
    pref = CO.roi_align(y.cpu().permute(0, 3, 1, 2).float(), torch.cat((bidx.float().unsqueeze(1), boxes), 1), 0.25)
    assert torch.allclose(pooled, pref, rtol=1e-4, atol=1e-5), "roi_align mismatch"
    scores = torch.tensor([[0.9, 0.8, 0.85]])
    keep, cnt = ops.nms_topk(boxes.view(1, 3, 4).to(dev), scores.to(dev), None, None, 1, 3, torch.tensor([3], dtype=torch.int32).to(dev), 0.5, 3)
    kref = CO.nms(boxes.numpy(), scores[0].numpy(), 0.5)
    assert keep[0, : int(cnt[0])].cpu().tolist() == kref.tolist(), "nms mismatch"
    torch.cuda.synchronize()
    print("smoke ok: conv err %.2e, roi_align ok, nms keep %s" % (err, kref.tolist()))
