import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GSSEG_LIB"] = os.path.join(root, "semantic_segmentation_amd", "libgsseg_hip_phase.so")
sys.path.insert(0, root)
import torch
from semantic_segmentation_amd import ops
dev = torch.device('cuda:0'); N = 32
names = ["sync even", "stage even(16b)", "sync odd", "stage odd(16b)", "epilogue", "item head", "stage even(fp8)", "stage odd(fp8)"]
for (H, cin, cout) in ((256, 64, 64), (128, 256, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    yh = torch.randn(N, H, H, cin, device=dev, generator=g).abs().half(); yl = (torch.randn(N, H, H, cin, device=dev, generator=g) * 1e-4).half()
    one = torch.ones(cin, device=dev); zero = torch.zeros(cin, device=dev)
    xq = torch.empty(N, H, H, 2 * cin, dtype=torch.float16, device=dev)
    ops.bn_act_apply_split_q8(yh, yl, one, zero, 0, xq, xq[..., cin:], True, 2 * cin, 0)
    xp = torch.empty(N, H, H, 2 * cin, dtype=torch.float16, device=dev)
    ops.bn_act_apply_split(yh, yl, one, zero, 0, xp, xp[..., cin:], 2 * cin, 0)
    wt = ((torch.rand(cout, cin, 3, 3, device=dev, generator=g) * 2 - 1) / (cin * 9) ** 0.5).contiguous()
    qpack = torch.empty(9, cout, 2 * cin, dtype=torch.float16, device=dev); wexp = torch.empty(cout, dtype=torch.int32, device=dev)
    ops.pack_weight_q8([(wt, qpack, wexp)])
    spack = torch.empty(9, cout, 3 * cin, dtype=torch.float16, device=dev)
    ops.pack_weight_segs([(wt, spack, False, [(0, 0, cin), (0, 0, cin), (1, 0, cin)])])
    y_hi = torch.empty(N, H, H, cout, dtype=torch.float16, device=dev); y_lo = torch.empty_like(y_hi)
    for form in (8, 4):
        ops.conv3x3_set_kernel_form(form)
        for kind in ("xw", "q"):
            part = torch.zeros(max(ops.bn_partials_numel(ops.conv3x3_mtiles(N, H, H, cout), cout), 301 * 2 * cout + 64), device=dev)
            for _ in range(3):
                if kind == "q":
                    ops.conv3x3_q8(xq, qpack, wexp, y_hi, y_lo, N, H, H, cin, cout, 2 * cin, 0, part)
                else:
                    ops.conv3x3_segs(xp, spack, y_hi, y_lo, N, H, H, 3 * cin, 2 * cin, cin, cout, in_stride=2 * cin, bn_partials=part)
            torch.cuda.synchronize()
            allw = part[300 * 2 * cout: 300 * 2 * cout + 64].cpu().view(8, 8)
            for w in ((0, 5) if form == 8 else (0, 2)):
                ph = allw[w].tolist(); tot = sum(ph)
                print(f"{cin}->{cout}@{H} {form}-wave {kind:2s} wave{w} cycles {int(tot):9d}  " + "  ".join(f"{n} {int(v):8d}" for n, v in zip(names, ph) if v), flush=True)
    ops.conv3x3_set_kernel_form(-1)
