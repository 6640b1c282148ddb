"""DESIGN.md §9 "fewer FLOPs instead of faster FLOPs": what a Winograd F(2,3) form of the vocoder's dilated convs would do to the
rounding, measured on CPU in float32 against float64 (no kernel exists; this only prices the idea).

A k-tap dilated conv is split into ceil(k / 3) groups of 3 taps (zero-padded); each group is an F(2,3) minimal-filtering problem on
the dilation lattice (outputs t and t + d from inputs t - d, t, t + d, t + 2 d relative to the group's first tap), and the four
transform-domain products are ACCUMULATED over groups and input channels before one output transform -- 4 multiplies per output pair
and group instead of 6: 1.5x fewer MACs at k = 3, 1.17x at k = 7 (9 taps), 1.375x at k = 11 (12 taps).
    python tools/winograd_error.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def direct(x, w, dil, dtype):
    """x [C, T], w [Co, Ci, k]; 'same' dilated conv."""
    k = w.shape[2]
    pad = dil * (k - 1) // 2
    return F.conv1d(x.to(dtype)[None], w.to(dtype), dilation=dil, padding=pad)[0]


def winograd23(x, w, dil):
    """float32 F(2,3) on the dilation lattice, transform-domain accumulation over tap groups and channels."""
    Co, Ci, k = w.shape
    pad = dil * (k - 1) // 2
    T = x.shape[1]
    ng = (k + 2) // 3
    wp = F.pad(w, (0, 3 * ng - k))                                  # zero taps at the end
    # output pairs (t, t + d): t runs over residues r < d of blocks of 2 d
    Tp = ((T + 2 * dil - 1) // (2 * dil)) * 2 * dil
    xp = F.pad(x, (pad, Tp - T + dil * (3 * ng - 1) - pad + 2 * dil))
    base = torch.arange(Tp).view(-1, 2, dil)[:, 0, :].reshape(-1)    # first output of every pair
    m = [torch.zeros(Co, base.numel()) for _ in range(4)]
    for g in range(ng):
        g0, g1, g2 = wp[:, :, 3 * g], wp[:, :, 3 * g + 1], wp[:, :, 3 * g + 2]
        G = [g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2]     # filter transform (done once per layer at pack time)
        off = 3 * g * dil
        d0, d1, d2, d3 = (xp[:, base + off + j * dil] for j in range(4))
        D = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]                     # input transform (one VALU op per fragment)
        for f in range(4):
            m[f] += G[f] @ D[f]
    y0, y1 = m[0] + m[1] + m[2], m[1] - m[2] - m[3]                  # output transform
    y = torch.zeros(Co, Tp)
    y[:, base] = y0
    y[:, base + dil] = y1
    return y[:, :T]


def rel(a, b):
    return float(torch.sqrt(torch.mean((a.double() - b.double()) ** 2)) / torch.sqrt(torch.mean(b.double() ** 2)))


def main():
    C, T = 64, 4000
    print(f"{C} channels, {T} rows; relative RMS error against float64 of the direct conv")
    print("taps dil | direct float32 | Winograd F(2,3) float32 | MAC ratio")
    for k in (3, 7, 11):
        for dil in (1, 3, 5):
            x = F.leaky_relu(torch.randn(C, T), 0.1)
            w = torch.randn(C, C, k) * (C * k) ** -0.5
            ref = direct(x, w, dil, torch.float64)
            e_d = rel(direct(x, w, dil, torch.float32), ref)
            yw = winograd23(x, w, dil)
            e_w = rel(yw, ref)
            print(f"{k:4d} {dil:3d} | {e_d:.2e} | {e_w:.2e} | {k / (2.0 * ((k + 2) // 3)):.3f}x")
    # a whole MRF stage: three ResBlocks (k = 3 / 7 / 11), each three (dilated conv, k-conv at dilation 1) pairs with residuals, mean of the three
    x0 = torch.randn(C, T)
    Ws = {k: [(torch.randn(C, C, k) * (C * k) ** -0.5 * 0.5, torch.randn(C, C, k) * (C * k) ** -0.5 * 0.5) for _ in range(3)] for k in (3, 7, 11)}

    def stage(conv):
        outs = []
        for k in (3, 7, 11):
            x = x0.clone() if conv is not None else x0.double()
            for (w1, w2), dil in zip(Ws[k], (1, 3, 5)):
                if conv is None:
                    h = direct(F.leaky_relu(x, 0.1), w1, dil, torch.float64)
                    x = x + direct(F.leaky_relu(h, 0.1), w2, 1, torch.float64)
                else:
                    h = conv(F.leaky_relu(x, 0.1), w1, dil)
                    x = x + conv(F.leaky_relu(h, 0.1), w2, 1)
            outs.append(x)
        return (outs[0] + outs[1] + outs[2]) / 3.0

    ref = stage(None)
    e_d = rel(stage(lambda a, w, d: direct(a, w, d, torch.float32)), ref)
    e_w = rel(stage(winograd23), ref)
    print(f"one MRF stage (18 convs, residuals, mean): direct float32 {e_d:.2e}, Winograd F(2,3) float32 {e_w:.2e}")


if __name__ == "__main__":
    main()
