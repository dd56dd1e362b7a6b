"""Transformer encoder / conditional-cross-attention decoder on the HIP kernels.

Drop-in for the reference module of the same name (reference future_od/models/transformer.py):
same class names, constructor arguments and parameter names (so state_dicts interchange,
SURVEY.md 8b), different internals:

  * token tensors are BATCH-FIRST [batch, tokens, D] (= the NHWC backbone output, no transposes);
    positional inputs are batch-independent TABLES [tokens, D] and are broadcast inside kernels;
  * every Linear / attention / LayerNorm / element-wise step is a call into libfod_hip.so through
    future_od.native.functional (forward and backward);
  * the conditional cross-attention never concatenates [content | sine]: the attention kernel
    takes the two 32-wide parts of each head as separate operands (score = q1.k1 + q2.k2);
  * the single-key IMU attention is evaluated in its exact collapsed form (softmax over one key
    is 1, reference transformer.py:108-119 / SURVEY F7): one row per frame instead of N rows.

Dropout: the reference applies dropout(0.1) in training mode.  These modules implement the
eval-mode graph (dropout = identity) in both modes for now -- see DESIGN.md "Gaps".
"""
import math
import os
from typing import List, Optional

import torch
from torch import Tensor, nn

from future_od.native import functional as Fn


def _reset_parameters(parameters):
    for p in parameters:
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


def _lin(x, m: nn.Linear, relu=False, out_f32=False):
    return Fn.linear(x, m.weight, m.bias, relu=relu, out_f32=out_f32)


class _Proj:
    """An output projection that has not been applied yet: `_add_norm` applies it together with the residual add and
    the post-norm as one autograd node (Fn.linear_add_norm), or on its own when dropout sits in between."""
    __slots__ = ("a", "lin", "a_relu")

    def __init__(self, a, lin, a_relu=False):
        self.a, self.lin, self.a_relu = a, lin, a_relu      # a_relu: see Fn.LinearAddNormFn / _fused_tail


def _add_norm(x, new, norm: nn.LayerNorm, p, training, then: nn.Linear = None):
    """norm(x + dropout(new)) -- the post-norm residual step of every sub-layer (reference transformer.py:271-272,
    285-286, 310-311, 417-418); `new` is a tensor or a pending `_Proj`.  `then`: the Linear the NEXT block applies to this
    result first (a cross-attention block's query_content): where the fused launch can, it computes that projection too
    and the return value is (y, then(y)) -- else (y, None)."""
    if isinstance(new, _Proj):
        vec_ok = new.lin.weight.shape[0] % 8 == 0
        if vec_ok and not (training and p > 0.0):
            if then is not None:
                if not new.a_relu and Fn.linear_add_norm_then_fits(new.a, new.lin.weight, then.weight):
                    return Fn.linear_add_norm_then(new.a, x, new.lin.weight, new.lin.bias, norm.weight, norm.bias,
                                                   then.weight, then.bias)
                return Fn.linear_add_norm(new.a, x, new.lin.weight, new.lin.bias, norm.weight, norm.bias, new.a_relu), None
            return Fn.linear_add_norm(new.a, x, new.lin.weight, new.lin.bias, norm.weight, norm.bias, new.a_relu)
        assert not new.a_relu, "a ReLU that left its gradient mask to a fused tail needs that tail (see _fused_tail)"
        new = _lin(new.a, new.lin)
    y = Fn.layer_norm(x, norm.weight, norm.bias, residual=Fn.dropout(new, p, training))
    return (y, None) if then is not None else y


def _fused_tail(lin: nn.Linear, p, training):
    """Whether `_add_norm` will apply `lin` inside the fused projection + add + norm node (its own condition): only then
    may the feed-forward's ReLU leave the masking of its gradient to that node."""
    return lin.weight.shape[0] % 8 == 0 and not (training and p > 0.0)


class MLP(nn.Module):
    """Linear -> ReLU -> ... -> Linear (reference transformer.py:18-32)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x, out_f32=False):
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            x = _lin(x, layer, relu=not last, out_f32=out_f32 and last)
        return x


class OutProj(nn.Module):
    """Parameter holder named like ConditionalDETR's MultiheadAttention (`fun.out_proj.*`)."""

    def __init__(self, vdim):
        super().__init__()
        self.out_proj = nn.Linear(vdim, vdim)
        nn.init.constant_(self.out_proj.bias, 0.0)


def _no_masks(attn_mask, key_padding_mask):
    """The reference's wrappers hand attn_mask / key_padding_mask on to MultiheadAttention (transformer.py:61-82,
    122-181, 401-419) and every caller in the reference passes None (transformer.py:270-307, 464-487, paper.py:164):
    the attention kernels have no mask operand, so anything else is refused rather than ignored."""
    if attn_mask is not None or key_padding_mask is not None:
        raise NotImplementedError("attention masks: fod_attn_fwd has no mask operand (the reference always passes None)")


class Attention(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.query_content = nn.Linear(D, D)
        self.query_pos = nn.Linear(D, D)
        self.key_content = nn.Linear(D, D)
        self.key_pos = nn.Linear(D, D)
        self.value = nn.Linear(D, D)


class SlotToSlotAttention(Attention):
    """Query self-attention (reference transformer.py:61-82).  x [B,M,D], qpos table [M,D]."""

    def __init__(self, D, Nhead, dropout):
        super().__init__(D)
        self.fun = OutProj(D)
        self.Nhead, self.D = Nhead, D
        self.droprate = dropout            # on the attention probabilities (MultiheadAttention(dropout=...), reference :64)

    def forward(self, x, qpos, pos_proj=None, attn_mask=None, key_padding_mask=None):
        """`pos_proj` = (query_pos(qpos), key_pos(qpos)) when the decoder has projected the (layer-independent)
        query positions for all layers at once."""
        _no_masks(attn_mask, key_padding_mask)
        M = qpos.shape[0]
        qp, kp = pos_proj if pos_proj is not None else (_lin(qpos, self.query_pos), _lin(qpos, self.key_pos))
        # q = q_content + q_pos, k = k_content + k_pos (positions: [M, D] tables shared by the batch) from the projections' launch
        q, k, v = Fn.group_linear(x, [self.query_content, self.key_content, self.value], residual=[qp, kp], res_row_mod=M)
        a = Fn.attention(q, k, v, 1.0 / math.sqrt(self.D // self.Nhead), drop_p=self.droprate, training=self.training)
        return _Proj(a, self.fun.out_proj)

    def forward_cross(self, x, qpos, other):
        """Queries attend to another query set `other` [B,M,D] that shares their positions (the previous frame's
        final queries, reference transformer.py:288-296 with key_pos = query_pos, :368)."""
        M = qpos.shape[0]
        qc = _lin(x, self.query_content)
        kc, v = Fn.group_linear(other, [self.key_content, self.value])
        q = Fn.add(qc, _lin(qpos, self.query_pos), b_row_mod=M)
        k = Fn.add(kc, _lin(qpos, self.key_pos), b_row_mod=M)
        a = Fn.attention(q, k, v, 1.0 / math.sqrt(self.D // self.Nhead), drop_p=self.droprate, training=self.training)
        return _Proj(a, self.fun.out_proj)


# TRAIN mode with active dropout and ONE IMU token per frame: the reference (transformer.py:108-119, 444, 485) draws its
# dropout masks per TOKEN -- on the attention probability (a dropped weight zeroes that token's whole IMU contribution),
# on the block's output and inside its MLP.  The collapsed one-key form below computes one row per frame, so its masks
# would be per frame; by default train mode therefore runs the general form (keys = the one token) and is mask-for-mask
# the reference's computation.  FOD_IMU_COLLAPSED_TRAIN=1 keeps the collapsed form in train mode too (faster, per-frame
# masks).  Eval mode (the benchmark, and every parity fixture) is unaffected: without dropout the two forms are equal.
IMU_COLLAPSED_TRAIN = os.environ.get("FOD_IMU_COLLAPSED_TRAIN", "0") == "1"


class EgodeepAttention(nn.Module):
    """Attention to the per-frame IMU token(s) (reference transformer.py:85-119).

    With ONE key per frame the softmax weight is exactly 1, so the output is
    out_proj(value(ego)) for every query row, and the following LayerNorm/MLP act on identical
    rows: computed once per frame as [frames, D]; the caller broadcasts.  query_content /
    query_pos / key only feed the (constant) softmax: their gradient is exactly zero."""

    def per_token_masks(self):
        """True when the one-key case must run the general form: dropout is active and its masks are per token."""
        return self.training and self.droprate > 0.0 and not IMU_COLLAPSED_TRAIN

    def __init__(self, D, Nhead, droprate, Dff=None):
        super().__init__()
        self.query_content = nn.Linear(D, D)
        self.query_pos = nn.Linear(D, D)
        self.key = nn.Linear(D, D)
        self.value = nn.Linear(D, D)
        self.fun = OutProj(D)
        self.use_mlp = Dff is not None
        self.droprate = droprate
        self.Nhead = Nhead
        if self.use_mlp:
            self.norm1 = nn.LayerNorm(D)
            self.mlp = nn.Sequential(nn.Linear(D, Dff), nn.ReLU(inplace=True), nn.Dropout(droprate),
                                     nn.Linear(Dff, D), nn.Dropout(droprate))
            self.norm2 = nn.LayerNorm(D)

    def dead_parameters(self):
        return [p for m in (self.query_content, self.query_pos, self.key) for p in m.parameters()]

    def forward_keys(self, x, pos, keys):
        """The general form: x [B,T,D] attends to `keys` [B,S,D] (the IMU tokens of S frames; reference :108-119 with
        S > 1, which only JointEncoder / the all-at-once detector produce).  pos: [T,D] table or [B,T,D]."""
        T_, D = x.shape[-2], x.shape[-1]
        qc = _lin(x, self.query_content)
        qp = _lin(pos, self.query_pos)
        q = Fn.add(qc, qp, b_row_mod=T_) if pos.dim() == 2 else Fn.add(qc, qp)
        k, v = Fn.group_linear(keys.contiguous(), [self.key, self.value])
        a = Fn.attention(q, k, v, 1.0 / math.sqrt(D // self.Nhead), drop_p=self.droprate, training=self.training)
        return self._tail(_lin(a, self.fun.out_proj))

    def forward_single_key(self, ego):
        """ego [frames, D] -> [frames, D]."""
        return self._tail(_lin(_lin(ego, self.value), self.fun.out_proj))

    def _tail(self, out):
        if self.use_mlp:
            # train mode: the reference drops per token; on the collapsed rows the masks are per frame
            t, p = self.training, self.droprate
            out = Fn.layer_norm(out, self.norm1.weight, self.norm1.bias, residual=Fn.dropout(out, p, t))   # norm1(out + drop(out))
            h = Fn.dropout(_lin(out, self.mlp[0], relu=True), p, t)
            out = Fn.layer_norm(out, self.norm2.weight, self.norm2.bias,
                                residual=Fn.dropout(_lin(h, self.mlp[3]), p, t))
        return out


class SlotToImageAttention(Attention):
    """Conditional cross-attention (reference transformer.py:122-181).

    x [B,M,D]; qpos table [M,D]; query_sine [B,M,D] or table [M,D]; mem [B,N,D]; mem_pos table [N,D]."""

    def __init__(self, D, Nhead, dropout):
        super().__init__(D)
        self.query_sine = nn.Linear(D, D)
        self.fun = OutProj(D)
        self.D, self.Nhead = D, Nhead
        self.droprate = dropout            # on the attention probabilities (reference :126)
        self.store_attention = False

    def forward(self, x, qpos, query_sine, side, layer, image, is_first, qs=None, qpos_proj=None, keep=False,
                attn_mask=None, key_padding_mask=None, qc_pre=None):
        """`side` (functional.MemorySide) holds this image's value / key_content projections for ALL layers
        and the projected positional table; this call uses the (layer, image) slots.  `qs` = query_sine(query_sine)
        and `qpos_proj` = query_pos(qpos) when the caller has projected them for all images at once.
        `keep`: also return x for the residual step (one autograd consumer of x instead of two, Fn.LinearKeepFn)."""
        _no_masks(attn_mask, key_padding_mask)
        B, M, D = x.shape
        if keep and qc_pre is not None and Fn.FUSED_LINEAR_PRE:
            # query_content(x) is an output of the node that produced x (_add_norm(then=...)); so is its backward
            x_keep, qc = x, qc_pre.view(B, M, D)
        elif keep:
            # qc_pre: query_content(x) as the launch that produced x already computed it (_add_norm(then=...))
            x_keep, qc = Fn.linear_keep(x, self.query_content.weight, self.query_content.bias, precomputed=qc_pre)
        else:
            assert qc_pre is None
            qc = _lin(x, self.query_content)
        if is_first:
            qc = Fn.add(qc, qpos_proj if qpos_proj is not None else _lin(qpos, self.query_pos), b_row_mod=M)
        if qs is None:
            qs = _lin(query_sine, self.query_sine)
        if is_first and qs.dim() == 2:
            # (as below, with the broadcast of the batch-independent sine projection inside the same launch)
            qs = Fn.add(qc, qs, b_row_mod=M)
        elif qs.dim() == 2:
            qs = Fn.expand_rows(qs, B).view(B, M, D)
        elif is_first:
            # reference: k_content += k_sine in the first layer.  q_c.(k_c + k_s) + q_s.k_s == q_c.k_c + (q_c + q_s).k_s,
            # so the addition moves to the (tiny) query side and the key slots are used as they are
            qs = Fn.add(qs, qc)
        # per head: q = [qc | qs], k = [kc | ks] (64 wide), scale (2D/heads)^-0.5
        a = Fn.hoisted_cross_attention(qc, qs, side, layer, image, 1.0 / math.sqrt(2 * D // self.Nhead),
                                       drop_p=self.droprate, training=self.training)
        if self.store_attention:
            kc, ks, _v = side.slots(layer, image)
            self.stored_attention = _head_mean_weights(qc, kc, qs, ks.unsqueeze(0).expand(B, -1, -1), self.Nhead)
        return (_Proj(a, self.fun.out_proj), x_keep) if keep else _Proj(a, self.fun.out_proj)


@torch.no_grad()
def _head_mean_weights(qc, kc, qs, ks, H):
    """Visualisation only (demo.ipynb): head-averaged attention map [B,M,N]; off the hot path."""
    B, M, D = qc.shape
    N = kc.shape[1]
    hd = D // H
    q = torch.cat([qc.float().view(B, M, H, hd), qs.float().view(B, M, H, hd)], 3).transpose(1, 2)
    k = torch.cat([kc.float().view(B, N, H, hd), ks.float().view(B, N, H, hd)], 3).transpose(1, 2)
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(2 * hd), dim=-1)
    return w.mean(1)


class TransformerDecoderLayer(nn.Module):
    """self-attention -> one cross-attention per image -> FFN, post-norm (reference :184-312)."""

    def __init__(self, D, Nhead, Dff=2048, dropout=0.1, num_images=1, use_slotstates=False, use_egodeep=False):
        super().__init__()
        self.self_attend = SlotToSlotAttention(D, Nhead, dropout)
        self.norm_sa = nn.LayerNorm(D)
        self.image_attend = nn.ModuleList([SlotToImageAttention(D, Nhead, dropout) for _ in range(num_images)])
        self.norm_ia = nn.ModuleList([nn.LayerNorm(D) for _ in range(num_images)])
        if use_slotstates:                 # attention to the previous frame's final queries (reference :210-215)
            self.slotstates_attend = SlotToSlotAttention(D, Nhead, dropout)
            self.norm_ssa = nn.LayerNorm(D)
        else:
            self.slotstates_attend = None
        if use_egodeep:                    # attention to the frame's IMU token, no MLP (reference :217-222)
            self.egodeep_attend = EgodeepAttention(D, Nhead, droprate=dropout, Dff=None)
            self.norm_eda = nn.LayerNorm(D)
        else:
            self.egodeep_attend = None
        self.feedforward = nn.Sequential(nn.Linear(D, Dff), nn.ReLU(inplace=True), nn.Dropout(dropout),
                                         nn.Linear(Dff, D))
        self.norm_out = nn.LayerNorm(D)
        self.Nhead, self.D = Nhead, D
        self.droprate = dropout            # dropout_sa / dropout_ia / dropout_out and the feed-forward's (reference :201-234)

    def forward(self, x, qpos, query_sine, side, layer, is_first=False, pos_proj=None, slotstates=None, egodeep=None):
        """`pos_proj` (from TransformerDecoder): {"sa": (query_pos(qpos), key_pos(qpos)), "ca": [query_pos_i(qpos)]};
        slotstates [B,M,D] or None; egodeep [B,D] (ONE IMU token per sample) or None."""
        t, p = self.training, self.droprate
        o = self.self_attend(x, qpos, pos_proj["sa"] if pos_proj else None)
        # each post-norm launch also forms the NEXT cross-attention block's query_content projection of its output
        if side.K > 0:
            x, qc_pre = _add_norm(x, o, self.norm_sa, p, t, then=self.image_attend[0].query_content)
        else:
            x, qc_pre = _add_norm(x, o, self.norm_sa, p, t), None
        # the sine embedding is the same for every image of the layer: all their query_sine projections at once
        qs_all = Fn.group_linear(query_sine, [self.image_attend[i].query_sine for i in range(side.K)])
        for i in range(side.K):
            o, x = self.image_attend[i](x, qpos, query_sine, side, layer, i, is_first, qs=qs_all[i],
                                        qpos_proj=pos_proj["ca"][i] if (pos_proj and is_first) else None, keep=True,
                                        qc_pre=qc_pre)
            if i + 1 < side.K:
                x, qc_pre = _add_norm(x, o, self.norm_ia[i], p, t, then=self.image_attend[i + 1].query_content)
            else:
                x, qc_pre = _add_norm(x, o, self.norm_ia[i], p, t), None
        if self.slotstates_attend is not None and slotstates is not None:
            o = self.slotstates_attend.forward_cross(x, qpos, slotstates)
            x = _add_norm(x, o, self.norm_ssa, p, t)
        if self.egodeep_attend is not None and egodeep is not None and egodeep.dim() == 2 and self.egodeep_attend.per_token_masks():
            egodeep = egodeep.unsqueeze(1)             # train mode: the general form, dropout masks per token as in the reference
        if self.egodeep_attend is not None and egodeep is not None and egodeep.dim() == 3:     # [B,S,D]: several keys
            e = Fn.dropout(self.egodeep_attend.forward_keys(x, qpos, egodeep), p, t)
            x = Fn.layer_norm(x, self.norm_eda.weight, self.norm_eda.bias, residual=e)
        elif self.egodeep_attend is not None and egodeep is not None:
            # one key: the attention output is the same row for every query of a sample (see EgodeepAttention)
            e = Fn.dropout(self.egodeep_attend.forward_single_key(egodeep), p, t)
            x = Fn.layer_norm(x, self.norm_eda.weight, self.norm_eda.bias, residual=e, res_row_div=x.shape[1])
            if torch.is_grad_enabled():
                x = Fn.ZeroGradAnchor.apply(x, *self.egodeep_attend.dead_parameters())
        fuse = _fused_tail(self.feedforward[3], p, t)
        x, h = Fn.linear_keep(x, self.feedforward[0].weight, self.feedforward[0].bias, relu=True, grad_masked=fuse)
        h = Fn.dropout(h, p, t)
        return _add_norm(x, _Proj(h, self.feedforward[3], a_relu=fuse), self.norm_out, p, t)


class TransformerDecoder(nn.Module):
    """Reference transformer.py:315-398.  Returns (hs [layers,B,M,D], ref f32 [M,2])."""

    def __init__(self, layers, norm=None, return_intermediate=False, D=256):
        super().__init__()
        self.layers = layers
        for module in self.layers[1:]:          # only layer 0 adds the projected query positions
            for att in module.image_attend:
                att.query_pos.weight = None
                att.query_pos.bias = None
        self.norm = norm
        self.return_intermediate = return_intermediate
        self.D = D
        self.query_scale = MLP(D, D, D, 2)
        self.ref_point_head = MLP(D, D, 2, 2)
        _reset_parameters(self.parameters())

    def forward(self, x, qpos, mems, mem_poss, first_layer_special=True, slotstates=None, egodeep=None):
        """x [B,M,D] query content; qpos [M,D] learned query positions (shared by the batch); slotstates [B,M,D]
        (previous frame's final queries) or None; egodeep [B,D] or None."""
        B, M, D = x.shape
        ref, sine0 = Fn.RefPointSineFn.apply(self.ref_point_head(qpos), D)     # [M,2] f32, [M,D]
        # memory side, hoisted out of the layer loop: one GEMM per image for every layer's value /
        # key_content projection, one GEMM for every (layer, image) key_pos projection of the table
        K = len(mems)
        big = [Fn.wide_linear(mems[j], [m for layer in self.layers
                                        for m in (layer.image_attend[j].value, layer.image_attend[j].key_content)])
               for j in range(K)]
        ks_all = Fn.wide_linear(mem_poss[0], [layer.image_attend[j].key_pos for layer in self.layers for j in range(K)])
        side = Fn.MemorySide(big, ks_all, len(self.layers), K, D)
        # query side: every projection of the learned query positions (2 per layer for the self-attention, one
        # per image for the first layer's cross-attention) in one grouped launch
        pos_lin = [m for layer in self.layers for m in (layer.self_attend.query_pos, layer.self_attend.key_pos)]
        if first_layer_special:
            pos_lin += [self.layers[0].image_attend[j].query_pos for j in range(K)]
        pos_all = Fn.group_linear(qpos, pos_lin)
        inter = []
        sine_acc, first_scaled = Fn.TableGradAcc(), True
        for lid, layer in enumerate(self.layers):
            special = lid == 0 and first_layer_special
            pos_proj = {"sa": (pos_all[2 * lid], pos_all[2 * lid + 1]),
                        "ca": pos_all[2 * len(self.layers):] if special else None}
            if special:
                q_sine = sine0
            elif Fn.mlp2_mul_fits(x, self.query_scale, sine0):
                # the MLP and the product in one launch; the layers' gradients of sine0 meet in one buffer
                q_sine = Fn.mlp2_mul(x.view(B * M, D), self.query_scale, sine0, acc=sine_acc,
                                     hands_on=first_scaled).view(B, M, D)
                first_scaled = False
            else:
                q_sine = Fn.mul(self.query_scale(x).view(B * M, D), sine0, b_row_mod=M).view(B, M, D)
            x = layer(x, qpos, q_sine, side, lid, is_first=special, pos_proj=pos_proj, slotstates=slotstates,
                      egodeep=egodeep)
            if self.return_intermediate:
                inter.append(x)
        if not self.return_intermediate:
            y = Fn.layer_norm(x, self.norm.weight, self.norm.bias) if self.norm is not None else x
            return y.unsqueeze(0), ref
        # the shared output norm over all levels at once (rows are independent: the same numbers as level by level):
        # one launch forward and one backward instead of six each, and one gradient for its weight / bias instead of six
        # that autograd would sum with ten kernels of its own
        return Fn.layer_norm(torch.stack(inter), self.norm.weight, self.norm.bias), ref


class PackedMHA(nn.Module):
    """Parameter holder with torch.nn.MultiheadAttention's names (in_proj_weight, in_proj_bias, out_proj)."""

    def __init__(self, D):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * D, D))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * D))
        self.out_proj = nn.Linear(D, D)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class EncoderAttention(nn.Module):
    """Post-norm self-attention + FFN block (reference transformer.py:401-419)."""

    def __init__(self, Dsrc, num_heads, Dff, droprate=0.1):
        super().__init__()
        self.attn = PackedMHA(Dsrc)
        self.norm1 = nn.LayerNorm(Dsrc)
        self.mlp = nn.Sequential(nn.Linear(Dsrc, Dff), nn.ReLU(inplace=True), nn.Dropout(droprate),
                                 nn.Linear(Dff, Dsrc), nn.Dropout(droprate))
        self.norm2 = nn.LayerNorm(Dsrc)
        self.D, self.H = Dsrc, num_heads
        self.droprate = droprate

    def forward(self, src, pos, other=None, attn_mask=None, key_padding_mask=None):
        """src [F,N,D]; pos table [N,D] or per-batch [F,N,D].  Self form: q = k = src + pos, v = src.  Cross form
        (`other` [F,N,D]: the previous frame's output or an earlier frame's features, reference :464-478):
        q = src + pos, k = other + pos, v = other."""
        _no_masks(attn_mask, key_padding_mask)
        D = self.D
        N = pos.shape[-2]
        add_pos = (lambda t: Fn.add(t, pos, b_row_mod=N)) if pos.dim() == 2 else (lambda t: Fn.add(t, pos))
        fused = Fn.in_proj_self(src, pos, self.attn.in_proj_weight, self.attn.in_proj_bias) if other is None else None
        if fused is not None:
            src, q, k, v = fused               # one autograd consumer of src (positional add, projections, residual)
        elif other is None:
            q, k, v = Fn.in_proj(add_pos(src), src, self.attn.in_proj_weight, self.attn.in_proj_bias)
        else:
            q, k, v = Fn.in_proj_cross(add_pos(src), add_pos(other), other, self.attn.in_proj_weight,
                                       self.attn.in_proj_bias)
        a = Fn.attention(q, k, v, 1.0 / math.sqrt(D // self.H), drop_p=self.droprate, training=self.training)
        t, p = self.training, self.droprate
        src = _add_norm(src, _Proj(a, self.attn.out_proj), self.norm1, p, t)
        fuse = _fused_tail(self.mlp[3], p, t)
        src, h = Fn.linear_keep(src, self.mlp[0].weight, self.mlp[0].bias, relu=True, grad_masked=fuse)
        h = Fn.dropout(h, p, t)
        return _add_norm(src, _Proj(h, self.mlp[3], a_relu=fuse), self.norm2, p, t)


class TransformerEncoderLayer(nn.Module):
    """Reference transformer.py:422-487.  `use_prevout` / `num_previmages` add cross-attention blocks onto the
    previous frame's output / earlier frames' features (only JointEncoderSequential feeds them)."""

    def __init__(self, D, Nhead, Dff=2048, droprate=0.1, num_previmages=0, use_prevout=False, use_egodeep=False):
        super().__init__()
        self.self_attn = EncoderAttention(D, Nhead, Dff, droprate=droprate)
        self.prevout_attn = EncoderAttention(D, Nhead, Dff, droprate=droprate) if use_prevout else None
        self.previmage_attn = nn.ModuleList(
            [EncoderAttention(D, Nhead, Dff, droprate=droprate) for _ in range(num_previmages)])
        if use_egodeep:
            self.egodeep_attend = EgodeepAttention(D, Nhead, droprate=droprate, Dff=Dff)
            self.norm_eda = nn.LayerNorm(D)
        else:
            self.egodeep_attend = None

    def forward(self, x, pos, egodeep: Optional[Tensor] = None, prevout: Optional[Tensor] = None, memory=None,
                imu_pre: Optional[Tensor] = None, imu_queue=None):
        """x [F,N,D]; pos table [N,D] or [F,N,D]; egodeep [F,D] (one IMU token per frame), [F,S,D] (S tokens) or
        None; prevout [F,N,D] or None; memory: list of [F,N,D] (most recent first) or None.  `imu_pre` [F,D]: this
        layer's one-key IMU block as TransformerEncoder computed it for all layers at once."""
        x = self.self_attn(x, pos)
        if prevout is not None and self.prevout_attn is not None:
            x = self.prevout_attn(x, pos, other=prevout)
        if memory is not None:
            for prev, attn in zip(memory, self.previmage_attn):
                x = attn(x, pos, other=prev)
        if egodeep is not None and self.egodeep_attend is not None and egodeep.dim() == 2 and self.egodeep_attend.per_token_masks():
            egodeep = egodeep.unsqueeze(1)             # train mode: the general form, dropout masks per token as in the reference
        if egodeep is not None and self.egodeep_attend is not None and egodeep.dim() == 3:   # [F,S,D]: several keys
            e = self.egodeep_attend.forward_keys(x, pos, egodeep)
            e = Fn.dropout(e, self.egodeep_attend.droprate, self.training)
            x = Fn.layer_norm(x, self.norm_eda.weight, self.norm_eda.bias, residual=e)
        elif egodeep is not None and self.egodeep_attend is not None:
            N = x.shape[1]
            e = imu_pre if imu_pre is not None else self.egodeep_attend.forward_single_key(egodeep)
            e = Fn.dropout(e, self.egodeep_attend.droprate, self.training)          # dropout_eda (reference :444,485)
            if imu_queue is not None and e is imu_pre:
                # (queue, slot): the gradient of this layer's IMU rows is summed with the other layers' in one launch
                x = Fn.layer_norm(x, self.norm_eda.weight, self.norm_eda.bias, residual=e, res_row_div=N,
                                  res_queue=imu_queue[0], res_slot=imu_queue[1])
            else:
                x = Fn.layer_norm(x, self.norm_eda.weight, self.norm_eda.bias, residual=e, res_row_div=N)
            if torch.is_grad_enabled():
                x = Fn.ZeroGradAnchor.apply(x, *self.egodeep_attend.dead_parameters())
        return x


class TransformerEncoder(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = layers
        _reset_parameters(self.parameters())

    def forward(self, x, pos, egodeep: Optional[Tensor] = None, prevout: Optional[Tensor] = None, memory=None):
        # one IMU token per frame: every layer's IMU block reads only that token (softmax over one key is 1), so the
        # blocks of all layers are computed here at once -- a handful of launches instead of 6 per layer (Fn.ImuBranchFn)
        pre = None
        blocks = [getattr(layer, "egodeep_attend", None) for layer in self.layers]
        if (egodeep is not None and egodeep.dim() == 2 and all(b is not None for b in blocks)
                and not any(b.training and b.droprate > 0.0 for b in blocks) and Fn.imu_branch_fits(egodeep, blocks)):
            queue = Fn.ResGradQueue(len(blocks)) if (Fn.RES_GRAD_QUEUE and torch.is_grad_enabled()) else None
            pre = Fn.imu_branch(egodeep, blocks, queue=queue)
        for i, layer in enumerate(self.layers):
            if pre is not None:
                x = layer(x, pos, egodeep, prevout, memory, imu_pre=pre[i],
                          imu_queue=None if queue is None else (queue, i))
            else:
                x = layer(x, pos, egodeep, prevout, memory)
        return x
