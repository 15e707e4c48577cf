import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epn_pointcloud_amd import ops, _lib, schedule as S
from epn_pointcloud_amd.vgtk import pc as pctk, so3conv as sptk
lib = _lib.get_lib()
gpu = torch.device("cuda:0")
l = S.cls_so3net_schedule(1024)[1]
for b, n in ((1, 512), (2, 512), (4, 512), (8, 512), (16, 512), (32, 512), (32, 256), (32, 128)):
    torch.manual_seed(1)
    pts = S.synthetic_clouds(b, n, gpu, seed=5)
    xyz = pts.permute(0, 2, 1).contiguous()
    conv = sptk.InterSO3Conv(l.cin, l.cout, 1, l.stride, l.radius, l.sigma, l.nn, lazy_sample=True).to(gpu)
    idx = pctk.ball_query_index(xyz, xyz, l.radius, l.nn)
    geo = ops.InterGeometry(xyz, xyz, idx, conv.anchors, conv.kernels, l.sigma)
    feats = torch.randn(b, l.cin, n, 60, device=gpu).contiguous(memory_format=torch.channels_last)
    d = geo.desc(l.cin, l.cout)
    cols = b * n * 60
    ck = l.cin * 24
    ws = torch.empty(int(lib.epn_inter_group_workspace_bytes(ctypes.byref(d))) + 16, dtype=torch.uint8, device=gpu)
    Gp = torch.full((cols, ck), 7.0, device=gpu)
    Gq = torch.full((cols, ck), 7.0, device=gpu)
    _lib.check(lib.epn_inter_group_packed_f32(ctypes.byref(d), feats.data_ptr(), Gp.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_of(feats)), "p")
    _lib.check(lib.epn_inter_group_f32(ctypes.byref(d), feats.data_ptr(), Gq.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_of(feats)), "q")
    pos = torch.empty(ck, dtype=torch.int32)
    lib.epn_inter_packed_position(l.cin, 24, ctypes.c_void_p(pos.data_ptr()))
    Gq2 = torch.empty_like(Gq)
    Gq2[:, pos.long().to(gpu)] = Gq
    diff = (Gp - Gq2).abs()
    bad_rows = (diff.max(dim=1).values > 1e-5).nonzero().flatten()
    print(b, n, "max diff", diff.max().item(), "bad rows", bad_rows.numel(), bad_rows[:8].tolist(), "untouched", int((Gp == 7.0).sum()))
    if b == 1 and bad_rows.numel() > 1:
        r = bad_rows[0].item()
        bad = (diff[r] > 1e-5).nonzero().flatten()
        print(" row", r, "bad elems", bad.numel(), bad[:40].tolist())
        print(" got", Gp[r, bad[:8]].tolist(), "want", Gq2[r, bad[:8]].tolist())
        r2 = bad_rows[1].item()
        bad2 = (diff[r2] > 1e-5).nonzero().flatten()
        print(" row", r2, "bad elems", bad2.numel(), bad2[:40].tolist())
        # is the wrong row equal to some other row's correct content?
        cand = (Gq2 - Gp[r]).abs().max(dim=1).values
        print(" closest row to got:", cand.argmin().item(), cand.min().item())
    if b == 1 and bad_rows.numel() > 1:
        r = bad_rows[0].item()
        bad = (diff[r] > 1e-5).nonzero().flatten()
        for pidx in bad[:4].tolist():
            v = Gp[r, pidx]
            hit = (Gq2[max(0, r - 40):r + 40] == v).nonzero()
            print("  bad value at", r, pidx, "found in correct tensor at", [(max(0, r - 40) + h[0].item(), h[1].item()) for h in hit[:4]])
