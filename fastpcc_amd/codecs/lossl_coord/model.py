"""Float twin of the integer LiDAR codec and its post-training quantisation: module tree, parameter names and bitstream of
/root/reference/models/convolutional/lossl_coord/model.py (predictors :28-274, codec :507-631, PTQ hooks :633-642,
observer insertion :685-722, float -> integer module replacement :725-888).

The float model exists to be calibrated and converted:

    model.pre_test_hook()            # residual blocks / sequences get histogram observers
    for frame in calibration_set:    # ordinary test passes feed the observers
        model(frame)
    model.post_test_hook()           # every float operator -> its fixed-point counterpart (import_parameters), state
                                     # dict written to cfg.int_param_save_path; loads into codecs.lossl_coord_int.Model

The traversal (which level is predicted from which, what is appended where) is shared with the integer model
(`codecs/lossl_coord_int/model.py`): the float predictors are subclasses that swap the operators and the encoding of the
occupancy bits (1.0 instead of 1 << 23).  Convolutions and linears run on fpcc_conv_f32 over the same lookup tables as the
integer operators; softmax -> 16-bit CDF is tensor arithmetic as in the reference (:465-472), the CDF rows go to the host
coder whole (this is the calibration path, not the product path).  `train_forward` (:394-424) runs the same network pass with
gradients: convolutions through fastpcc_amd/autograd.py, linears as library GEMMs."""
import io
import math
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hipops as ops
from ...int_sparse_conv import Conv3d, LinearIn8W8Out8, LinearIn8W8Out32, LinearPReLUIn8W8Out8, LinearPReLUIn8W8Out32, \
    PReLUIn32Out32, RequantFxpToScaledInt8, SparseConvIn8W8Out8, SparseConvIn8W8Out32, SparseConvPReLUIn8W8Out8, \
    SparseConvPReLUIn8W8Out32, SparseResBlockIn32W8Out32, SparseResBlockWithObs, SparseTensor, \
    SparseTensorHistogramObserver, make_obs
from ..lossl_coord_int import model as int_model
from .model_config import Config

_ON_FEATURES_INT = (PReLUIn32Out32, RequantFxpToScaledInt8, LinearIn8W8Out8, LinearIn8W8Out32, LinearPReLUIn8W8Out8,
                    LinearPReLUIn8W8Out32)


def linear(x: torch.Tensor, m: nn.Linear) -> torch.Tensor:
    """x @ W^T + b on the convolution kernel (one offset, identity map): a row's result does not depend on the other rows"""
    if torch.is_grad_enabled() and (m.weight.requires_grad or x.requires_grad):
        return F.linear(x.float(), m.weight, m.bias)                    # training: a plain library GEMM with autograd
    w = m.weight.detach().t().contiguous().reshape(1, 1, m.in_features, m.out_features)
    return ops.conv_f32(x.float().contiguous(), w, m.out_features, x.shape[0], bias=None if m.bias is None else m.bias.detach())


class Block(nn.Module):
    """residual block (model.py:645-661)"""

    def __init__(self, ch: int):
        super().__init__()
        self.ch = ch
        self.conv = Conv3d(ch, ch, 3, 1, 1, bias=True)
        self.act = nn.PReLU()
        self.conv2 = Conv3d(ch, ch, 3, 1, 1, bias=True)
        self.act2 = nn.PReLU()

    def forward(self, org: SparseTensor) -> SparseTensor:
        x = self.conv(org)
        x.F = self.act(x.F)
        x = self.conv2(x)
        x.F = self.act2(x.F + org.F)
        return x


class SparseSequential(nn.Sequential):
    """dense modules act on the feature matrix, the others on the sparse tensor (model.py:664-673); unlike the reference's
    it also runs the integer operators, so a converted model can be evaluated in place"""

    def forward(self, input: SparseTensor) -> SparseTensor:
        x = SparseTensor(input.F, input.C, input.stride, input.spatial_range)
        x._caches = input._caches
        for module in self:
            if isinstance(module, nn.Linear):
                x.F = linear(x.F, module)
            elif isinstance(module, (nn.LayerNorm, nn.ReLU, nn.LeakyReLU, nn.PReLU)) or isinstance(module, _ON_FEATURES_INT):
                x.F = module(x.F)
            else:
                x = module(x)
        return x


class OneScalePredictor(int_model.OneScalePredictor):
    def __init__(self, channels, if_upsample=True, allow_single_ch=False):
        nn.Module.__init__(self)
        if allow_single_ch:
            self.dec_init = Conv3d(1, channels, 3, 1, 1, bias=True)
        self.dec = Block(channels)
        self.pred = SparseSequential(Conv3d(channels, channels, 3, 1, 1, bias=True), nn.PReLU(), nn.Linear(channels, 255, bias=True))
        self.if_upsample = if_upsample
        self.upsample = SparseSequential(nn.Linear(channels + 8, channels, bias=True), nn.PReLU(), Block(channels),
                                         nn.Linear(channels, channels * 8, bias=True)) if if_upsample else None

    @staticmethod
    def _feat(bits: torch.Tensor) -> torch.Tensor:
        return bits.to(torch.float32)


class OneScaleMultiStepPredictor(int_model.OneScaleMultiStepPredictor):
    _feat = staticmethod(OneScalePredictor._feat)

    def __init__(self, channels, pred_steps=2, use_more_ch_for_multi_step_pred=True):
        nn.Module.__init__(self)
        self.pred_steps = pred_steps
        span = 2 ** (pred_steps - 2)
        if pred_steps == 2:
            self.embed = SparseSequential()
            out_ch = channels
            self.dec = SparseSequential(nn.Linear(channels + 8, out_ch), nn.PReLU(), Block(out_ch))
        elif use_more_ch_for_multi_step_pred:
            if pred_steps == 3:
                emb, in_ch, out_ch = 64, channels + 64, round(channels * 1.25)
            elif pred_steps >= 4:
                emb, in_ch, out_ch = 512, round(channels * 1.25) + 512, channels * 2
            else:
                raise NotImplementedError
            self.embed = SparseSequential(Conv3d(8, emb, span, span, bias=True), nn.PReLU())
            self.dec = SparseSequential(nn.Linear(in_ch, out_ch), nn.PReLU(), Block(out_ch)) if in_ch != out_ch else Block(out_ch)
        else:
            if pred_steps < 3:
                raise ValueError(pred_steps)
            self.embed = SparseSequential(Conv3d(8, channels, span, span, bias=True))
            if channels >= 256:
                self.embed.append(nn.PReLU())
            self.dec = SparseSequential(nn.Linear(channels + channels, channels), nn.PReLU(), Block(channels))
            out_ch = channels
        self.pred = nn.ModuleList()
        for i in range(pred_steps):
            if i == 0:
                self.pred.append(SparseSequential(Conv3d(out_ch, out_ch, 3, 1, 1, bias=True), nn.PReLU(),
                                                  nn.Linear(out_ch, channels * 8, bias=True)))
            elif i != pred_steps - 1:
                self.pred.append(SparseSequential(nn.PReLU(), nn.Linear(channels + 8, channels, bias=True), nn.PReLU(),
                                                  Conv3d(channels, channels, 3, 1, 1, bias=True), nn.PReLU(),
                                                  nn.Linear(channels, channels * 8, bias=True)))
            else:
                self.pred.append(SparseSequential(Conv3d(channels, channels, 3, 1, 1, bias=True), nn.PReLU(),
                                                  nn.Linear(channels, 255, bias=True)))


class Model(int_model.Model):
    one_scale_cls, multi_step_cls = OneScalePredictor, OneScaleMultiStepPredictor

    def __init__(self, cfg: Config, device='cuda'):
        super().__init__(cfg, device)
        self.converted = False

    # -- float entropy parameters (model.py:465-472) -------------------------------------------------------------------
    @staticmethod
    def batch_quantize_pmf_torch(pmfs: torch.Tensor, softmax: bool = True) -> torch.Tensor:
        """[n, c] logits (or probabilities) -> uint16 CDF rows without the leading zero, every frequency >= 1"""
        if softmax:
            pmfs = F.softmax(pmfs.float(), dim=-1)
        pmfs = pmfs.mul(65536 - pmfs.shape[1]).floor_().add_(1)
        pmfs.cumsum_(-1)
        pmfs[:, -1] = 65535
        return pmfs.to(torch.int32)

    @staticmethod
    def _to_host_u16(t: torch.Tensor) -> np.ndarray:
        h = torch.empty(t.shape, dtype=torch.int32, pin_memory=t.is_cuda)
        h.copy_(t, non_blocking=True)
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        return h.numpy().astype(np.uint16)

    def rans_decode_oct(self, logits: torch.Tensor) -> torch.Tensor:
        if self.converted:
            return super().rans_decode_oct(logits)
        rows = self._to_host_u16(self.batch_quantize_pmf_torch(logits))
        out = np.empty(rows.shape[0], dtype=np.uint16)
        self.rans_decoder.decode(rows, out)
        symbols = torch.from_numpy(out.astype(np.int16)).to(logits.device)
        symbols._fpcc_children = int_model._children_count(out)
        return symbols

    def _ones(self, n: int, device) -> torch.Tensor:
        return torch.ones((n, 1), dtype=torch.int8 if self.converted else torch.float32, device=device)

    def get_bin(self, input: SparseTensor, ones_feats: torch.Tensor) -> SparseTensor:
        """occupancy bits come from the integer fold kernel (exact); the float predictors read them as 0.0 / 1.0"""
        ones8 = ones_feats if ones_feats.dtype == torch.int8 else torch.ones(ones_feats.shape, dtype=torch.int8, device=ones_feats.device)
        ret = super().get_bin(input, ones8)
        if not self.converted:
            ret.F = ret.F.to(torch.float32)
        return ret

    # -- codec (model.py:507-618) ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def compress(self, xyz: torch.Tensor) -> bytes:
        if self.converted:
            return super().compress(xyz)
        if not xyz.is_cuda:
            raise RuntimeError('compress() runs on the GPU; move the coordinates there first')
        coord_offset = xyz.amin(0)[1:]
        xyz = xyz - F.pad(coord_offset, (1, 0))
        _, perm = ops.sort_keys(ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)))
        xyz = xyz[perm.long()].contiguous()
        ones = self._ones(xyz.shape[0], xyz.device)
        org = SparseTensor(ones, xyz, (1, 1, 1))
        skip = self.cfg.skip_top_scales_num
        blocks = self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip
        strided = [org]
        for _ in range(levels):
            strided.append(self.get_bin(strided[-1], ones))
        top = strided[-1]
        bottom = top.C[:, 1:].reshape(-1)
        bottom_cdf = self.batch_quantize_pmf_torch((torch.bincount(bottom.to(torch.int32), minlength=2) / bottom.numel())[None], False)[0]
        cur_rec = SparseTensor(ones[:top.C.shape[0]], top.C, (2 ** levels,) * 3)
        cur_rec._caches = org._caches

        pending = []
        for idx in range(levels, 0, -1):
            block = self._block(idx, blocks)
            if isinstance(block, int_model.OneScalePredictor):
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx - 1], strided[idx].F, self.bin2oct_kernel,
                                                          if_upsample=idx != 1 and block.if_upsample)
            else:
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx: idx + block.pred_steps], self.bin2oct_kernel)
            pending.append((self._to_host_u16(self.batch_quantize_pmf_torch(logits)), self._to_host_u16(symbols.to(torch.int32))))
        while pending:                                           # finest level first: the decoder pops coarse -> fine
            rows, symbols = pending.pop()
            self.rans_encoder.encode(rows, symbols)
        bottom_h = self._to_host_u16(bottom.to(torch.int32))
        self.rans_encode_fea(self._to_host_u16(bottom_cdf), bottom_h)
        with io.BytesIO() as bs:
            for v in coord_offset.tolist():
                bs.write(int(v).to_bytes(2, 'little'))
            bs.write((bottom_h.shape[0] // 3).to_bytes(2, 'little'))
            bs.write(self.rans_encoder.flush())
            return bs.getvalue()

    def get_init_pc(self, xyz: torch.Tensor, stride: int = 1) -> SparseTensor:
        return SparseTensor(self._ones(xyz.shape[0], xyz.device), xyz, (stride,) * 3)

    # -- training (model.py:394-424) ---------------------------------------------------------------------------------------
    def forward(self, pc_data):
        if self.training:
            return self.train_forward(pc_data.xyz, pc_data.points_num, getattr(pc_data, 'training_step', 0))
        return super().forward(pc_data)

    def train_forward(self, xyz: torch.Tensor, points_num: List[int], training_step: int = 0) -> dict:
        """bits per input point of the 255-ary occupancy symbols of every level under the predicted distributions, averaged
        over the batch.  xyz int32 [N, 4], per sample Morton ('zyx') sorted and unique, samples in batch order.  The network
        pass is the encoder's (`compress` of the predictors) with gradients enabled; only the float model trains."""
        if self.converted:
            raise RuntimeError('the converted (integer) model does not train')
        batch_size = len(points_num)
        ones = self._ones(xyz.shape[0], xyz.device)
        org = SparseTensor(ones, xyz.contiguous(), (1, 1, 1))
        levels = self.max_downsample_times
        strided = [org]
        for _ in range(levels):
            strided.append(self.get_bin(strided[-1], ones))
        top = strided[-1]
        cur_rec = SparseTensor(ones[:top.C.shape[0]], top.C, (2 ** levels,) * 3)
        cur_rec._caches = org._caches
        per_sample = torch.tensor(points_num, dtype=torch.float32, device=xyz.device)
        losses = {}
        for idx in range(levels, 0, -1):
            block = self._block(idx, self.blocks_dec)
            if isinstance(block, int_model.OneScalePredictor):
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx - 1], strided[idx].F, self.bin2oct_kernel,
                                                          if_upsample=block.if_upsample)
            else:
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx: idx + block.pred_steps], self.bin2oct_kernel)
            per_row = per_sample[strided[idx].C[:, 0].long()]
            losses[f'stride{2 ** idx}_geo_loss'] = (F.cross_entropy(logits, symbols.long(), reduction='none') / per_row).sum() \
                * (math.log2(math.e) / batch_size)
        total = sum(losses.values())
        out = {k: v.item() for k, v in losses.items()}
        out['loss'] = total
        return out

    # -- post-training quantisation (model.py:633-642) -------------------------------------------------------------------
    def pre_test_hook(self):
        if self.cfg.quantize_param:
            insert_obs_into_resblocks(self)
            insert_obs_into_seqs(self)

    def post_test_hook(self):
        if self.cfg.quantize_param:
            for m in self.modules():                     # the parameter search is a Python loop over 2048 bins: host
                if isinstance(m, SparseTensorHistogramObserver):
                    m.cpu()
            replace_resblocks_with_int_impl(self)
            replace_seqs_with_int_impl(self)
            for m in self.modules():                     # occupancy bits are 1 << 23 from here on
                if isinstance(m, (OneScalePredictor, OneScaleMultiStepPredictor)):
                    m._feat = int_model.OneScalePredictor._feat
            self.converted = True
            torch.save({'state_dict': self.state_dict()}, self.cfg.int_param_save_path)


def _device_of(model: nn.Module):
    return next(model.parameters()).device


def insert_obs_into_resblocks(model: nn.Module):
    """every float residual block -> the same block with observers (model.py:685-698)"""
    device = _device_of(model)

    def walk(parent: nn.Module):
        for name, child in list(parent._modules.items()):
            if isinstance(child, Block):
                new = SparseResBlockWithObs(child.ch).to(device)
                new.load_state_dict(child.state_dict(), strict=False)
                parent._modules[name] = new
            else:
                walk(child)

    walk(model)


def insert_obs_into_seqs(model: nn.Module):
    """one observer in front of every member of a sequence: affine (asymmetric) in front of a linear, symmetric elsewhere
    (model.py:700-722)"""
    device = _device_of(model)
    scheme = lambda m: torch.per_tensor_affine if isinstance(m, nn.Linear) else torch.per_tensor_symmetric

    def walk(parent: nn.Module):
        for name, child in list(parent._modules.items()):
            if isinstance(child, SparseSequential) and len(child) > 0:
                members = [make_obs(scheme(child[0])).to(device)]
                for i, m in enumerate(child):
                    members.append(m)
                    if i < len(child) - 1:
                        members.append(make_obs(scheme(child[i + 1])).to(device))
                parent._modules[name] = SparseSequential(*members)
            else:
                walk(child)

    walk(model)


def replace_resblocks_with_int_impl(model: nn.Module):
    """(model.py:725-738)"""
    device = _device_of(model)

    def walk(parent: nn.Module):
        for name, child in list(parent._modules.items()):
            if isinstance(child, SparseResBlockWithObs):
                new = SparseResBlockIn32W8Out32(child.ch).to(device)
                new.import_parameters(child)
                parent._modules[name] = new
            else:
                walk(child)

    walk(model)


def replace_seqs_with_int_impl(model: nn.Module):
    """Observed float sequences -> fixed-point sequences (model.py:741-888).  Walking a sequence keeps one bit of state: are
    the activations scaled int8 (between two fused operators) or Q8.23 fixed point (everywhere else)?  A conv / linear
    absorbs the PReLU that follows it; it emits int8 iff another conv / linear comes next, with the scale of the observer that
    follows the fused pair; entering int8 from fixed point costs one requantiser with the scale of the observer in front."""
    device = _device_of(model)
    is_obs = lambda m: isinstance(m, SparseTensorHistogramObserver)

    def convert(seq: SparseSequential) -> SparseSequential:
        members = list(seq)

        def prev_obs(i):
            return next(members[k] for k in range(i - 1, -1, -1) if is_obs(members[k]))

        def next_where(i, pred):
            return next((k for k in range(i + 1, len(members)) if pred(members[k])), None)

        out, in_int8, i = [], False, 0
        while i < len(members):
            m = members[i]
            if is_obs(m):
                i += 1
            elif isinstance(m, (nn.Linear, Conv3d)):
                scale_in, zp_in = prev_obs(i).calculate_qparams()
                nxt = next_where(i, lambda t: not is_obs(t))
                prelu = members[nxt] if nxt is not None and isinstance(members[nxt], nn.PReLU) else None
                last = nxt if prelu is not None else i                      # last member of the fused pair
                after = next_where(last, lambda t: not is_obs(t))
                out_int8 = after is not None and isinstance(members[after], (nn.Linear, Conv3d))
                if not in_int8:
                    requant = RequantFxpToScaledInt8().to(device)
                    requant.import_parameters(scale_in, zp_in)
                    out.append(requant)
                scale_out, zp_out = members[next_where(last, is_obs)].calculate_qparams() if out_int8 else (None, None)
                if isinstance(m, nn.Linear):
                    cls = {(True, True): LinearPReLUIn8W8Out8, (True, False): LinearPReLUIn8W8Out32,
                           (False, True): LinearIn8W8Out8, (False, False): LinearIn8W8Out32}[(prelu is not None, out_int8)]
                    fused = cls(m.in_features, m.out_features).to(device)
                else:
                    # (the reference instantiates an Out32 operator for "conv, no PReLU, int8 out" and would fail on its
                    #  argument list, model.py:851-853; no model of the repository has that pattern)
                    cls = {(True, True): SparseConvPReLUIn8W8Out8, (True, False): SparseConvPReLUIn8W8Out32,
                           (False, True): SparseConvIn8W8Out8, (False, False): SparseConvIn8W8Out32}[(prelu is not None, out_int8)]
                    fused = cls(m.in_channels, m.out_channels, m.kernel_size, m.stride).to(device)
                args = [scale_in, zp_in] + ([scale_out, zp_out] if out_int8 else []) + [m] + ([prelu] if prelu is not None else [])
                fused.import_parameters(*args)
                out.append(fused)
                in_int8 = out_int8
                i = last + 1
            elif isinstance(m, nn.PReLU):
                if in_int8:
                    raise NotImplementedError('a stand-alone PReLU on scaled int8 activations')
                new = PReLUIn32Out32().to(device)
                new.import_parameters(m)
                out.append(new)
                i += 1
            elif isinstance(m, SparseResBlockIn32W8Out32):
                if in_int8:
                    raise NotImplementedError('a residual block on scaled int8 activations')
                out.append(m)
                i += 1
            else:
                raise NotImplementedError(m)
        return SparseSequential(*out)

    def walk(parent: nn.Module):
        for name, child in list(parent._modules.items()):
            if isinstance(child, SparseSequential):
                parent._modules[name] = convert(child)
            elif isinstance(child, Conv3d) and name.endswith('dec_init'):
                # the one convolution outside a sequence: its input is the all-ones feature, scale 1 (model.py:877-882)
                new = SparseConvIn8W8Out32(1, child.out_channels, child.kernel_size, child.stride).to(device)
                new.import_parameters(torch.ones(1, dtype=torch.float32, device=device),
                                      torch.zeros(1, dtype=torch.int32, device=device), child)
                parent._modules[name] = new
            else:
                walk(child)

    walk(model)
