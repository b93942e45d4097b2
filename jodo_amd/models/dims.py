"""Derived sizes of a DGT configuration (mirror of DgtDims in csrc/dgt_plan.h; reference ctor models/mol_gnn.py:417-439)."""


class ModelDims:
    """Derived sizes; mirrors DgtDims in csrc/dgt_plan.h."""

    def __init__(self, nf, n_layers, n_heads, n_extra, mlp_ratio, in_node_dim, edge_ch, cond_ch=0, wide=None):
        self.D, self.L, self.H, self.XH, self.r = nf, n_layers, n_heads, n_extra, mlp_ratio
        self.nd, self.ch, self.cond_ch = in_node_dim, edge_ch, cond_ch
        self.De, self.T = nf // 4, nf * 4
        self.SH = n_heads - n_extra
        self.C = nf // n_heads
        self.SC = (n_heads * self.C) // self.SH
        # wide = width-generic kernel set + its q/k arrangement (always for nf != 256; optional at 256)
        self.wide = bool(nf != 256) if wide is None else bool(wide or nf != 256)
        self.ndp = (2 * in_node_dim + 7) // 8 * 8
        self.einp = (2 * edge_ch + 7) // 8 * 8
        self.cn, self.ce = (2 * nf) // n_layers, (2 * self.De) // n_layers
        # padded per-block readout widths (mirror of dgt_dims_from_cfg): 64 / 16 at nf 256 (L >= 8), 96 / 32 at nf 384, 64 / 16 at nf 128 L 6
        self.cnp = max(nf // 4, (self.cn + 31) // 32 * 32)
        self.cep = max((nf // 16 + 15) // 16 * 16, (self.ce + 15) // 16 * 16)
        self.wide = self.wide or self.cnp != 64
        tail = self.SC - 16
        self.QKP = self.SH * 32 if self.wide else (self.SH // 2 + (tail + 1) // 2) * 32
        self.KNH, self.KEH = nf + n_layers * self.cnp, self.De + n_layers * self.cep
        # modulation slice of a block: node 6D | edge 6De | equi (shift, scale) 2D | gbf 2 (+30 pad) | coord_mlp.0 pushed
        # through the LayerNorm of equi_update: W0 (1 + scale) [D] | W0 shift + b0 [D]  (csrc/dgt_kernels_wide.h, pair update)
        self.MB = 6 * nf + 6 * self.De + 2 * nf + 32 + 2 * nf
        self.Mtot = 32 + n_layers * self.MB
