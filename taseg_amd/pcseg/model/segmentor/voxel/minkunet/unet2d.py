"""UNet2D - the camera branch of TIAF (reference pcseg/model/segmentor/voxel/minkunet/unet2d.py).

The dense 2-D convolutions / BatchNorm2d / pooling / PixelShuffle stay on PyTorch-ROCm (MIOpen / hipBLASLt are the
roofline for dense work - SURVEY.md section 2.1 row 4); module tree and parameter names are the reference's, so
its checkpoints load (`stem.0.conv1.weight`, `stage2.bn1.running_mean`, `up3.conv1.weight`, `classifier.0.bias`).

Memory format: `options.image_layout = "nhwc"` (default) keeps the whole branch in `torch.channels_last` - the layout MIOpen's
fp16 / NHWC solvers want (under `torch.autocast`, the mode the reference always trains in: dist_train.sh:18 `--amp`) and the layout
the reference itself indexes for the image -> point hand-over (unet2d.py:183-187: `permute(0, 2, 3, 1)` then row indexing);
"nchw" is the plain contiguous format.

What is rebuilt is that hand-over (unet2d.py:180-214): the reference materialises NHWC copies of five feature stacks and
fancy-indexes them per sample; here the FOV points are put in raster order of their pixels once per batch and scale
(`ts_image_plan`) and every map is gathered in place -
  * channels-last maps: a pixel is one contiguous row, moved in 16-byte pieces (`ts_image_gather_rows_forward`, any dtype); the
    adjoint sums the gradient rows of a pixel's points in index order and read-modify-writes the pixel's row once
    (`ts_image_gather_rows_backward`, fp32 / fp16 maps);
  * NCHW maps: lanes along the points of the raster order, one plane at a time, an LDS transpose into rows
    (`ts_image_gather_forward` / `_backward`, fp32);
both adjoints without atomics (run-to-run identical) and ADDED into the gradient the map already has from its dense consumer
(no zero-filled T x C x H x W stack).
"""
import torch
from torch import nn
from torch.autograd import Function

from taseg_amd import backend as B
from taseg_amd.options import options

__all__ = ["UNet2D", "image_gather", "image_plan"]


def _rows(feat):
    """is this map's MEMORY [T, hs, ws, C] (torch.channels_last)?  (a one-channel contiguous map is both)"""
    return feat.dim() == 4 and feat.is_contiguous(memory_format=torch.channels_last)


def _gather(feat, plan):
    if _rows(feat):
        return B.image_gather_rows_forward(feat, plan)
    return B.image_gather_forward(feat.contiguous().float(), plan)


class _ImageGather(Function):
    """rows of one map at the plan's pixels; the adjoint is a zero-filled stack + the per-pixel sums"""

    @staticmethod
    def forward(ctx, feat, plan):
        ctx.plan, ctx.channels, ctx.rows, ctx.dtype = plan, feat.shape[1], _rows(feat), feat.dtype
        return _gather(feat, plan)

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.rows and ctx.dtype in (torch.float32, torch.float16):
            return B.image_gather_rows_backward(grad_out, ctx.plan, ctx.channels, dtype=ctx.dtype), None
        return B.image_gather_backward(grad_out.contiguous().float(), ctx.plan, ctx.channels).to(ctx.dtype), None


class _ImageGatherThrough(Function):
    """(feat', rows) = (feat, rows of feat at the plan's pixels): the map leaves the node a second time, and whatever consumes
    THAT tensor (the next decoder stage, the classifier, the dense loss) sends its gradient back INTO the node, where the adjoint
    of the gather is added to it in place on the pixels that have points - no zero-filled T x C x H x W stack (1.9 GB for the
    96-channel full-resolution fp32 map of a bs-2 TIAF batch), no second dense add by the autograd engine.  Use feat' downstream.

    PRIVATE to UNet2D.forward: adding into the incoming gradient is only sound when that tensor is a buffer nobody else reads -
    a fresh result of the consumer's backward (Conv2d, PixelShuffle, the permute + reshape in front of the dense loss) or the sum
    the engine formed for this output.  A consumer such as `feat' + other` hands the SAME tensor object to both of its inputs; do
    not put one behind this node."""

    @staticmethod
    def forward(ctx, feat, plan):
        ctx.plan, ctx.shape, ctx.rows, ctx.dtype = plan, tuple(feat.shape), _rows(feat), feat.dtype
        return feat.view_as(feat), _gather(feat, plan)

    @staticmethod
    def backward(ctx, grad_feat, grad_rows):
        if grad_rows is None:
            return grad_feat, None
        c = ctx.shape[1]
        if ctx.rows and ctx.dtype in (torch.float32, torch.float16):
            if grad_feat is None:
                return B.image_gather_rows_backward(grad_rows, ctx.plan, c, dtype=ctx.dtype), None
            # (a gradient in another format / dtype is first brought into the map's: a copy, and then not the engine's buffer)
            into = grad_feat.to(ctx.dtype).contiguous(memory_format=torch.channels_last)
            return B.image_gather_rows_backward(grad_rows, ctx.plan, c, dtype=ctx.dtype, into=into), None
        g = grad_rows.contiguous().float()
        if grad_feat is None:
            return B.image_gather_backward(g, ctx.plan, c).to(ctx.dtype), None
        into = grad_feat if (grad_feat.dtype == torch.float32 and grad_feat.is_contiguous()) else grad_feat.float().contiguous()
        return B.image_gather_backward(g, ctx.plan, c, into=into).to(ctx.dtype), None


def image_plan(pix, pbatch, frame_end, frames, height, width, shift=0):
    """raster-order plan of the FOV points for one scale of the camera stack (backend.image_plan)"""
    return B.image_plan(pix, pbatch, frame_end, int(frames), int(height), int(width), int(shift))


def image_gather(feat, pix, pbatch, frame_end, height, width, shift=0, plan=None):
    """Rows of `feat` [T, C, H >> shift, W >> shift] at the pixels the FOV points project to -> ([n, C] of feat's dtype, err); a
    channels-last map is read as rows in place, any other as NCHW planes (fp32)."""
    if plan is None:
        plan = image_plan(pix, pbatch, frame_end, feat.shape[0], height, width, shift)
    return _ImageGather.apply(feat, plan), plan["err"]


def _image_gather_through(feat, plan):
    """(feat', rows): see _ImageGatherThrough"""
    return _ImageGatherThrough.apply(feat, plan)


class _AvgPool3s2Rows(Function):
    """AvgPool2d(3, stride 2, padding 1) on a channels-last map through csrc/image.hip (forward and gradient)"""

    @staticmethod
    def forward(ctx, x):
        ctx.in_shape, ctx.dtype = tuple(x.shape), x.dtype
        return B.avgpool3s2_rows_forward(x)

    @staticmethod
    def backward(ctx, grad_y):
        return B.avgpool3s2_rows_backward(grad_y.to(ctx.dtype), ctx.in_shape)


def _pool(module, x):
    """the block's AvgPool2d.  A channels-last map on the device goes through the library's own kernels: the gradient kernel this
    PyTorch-ROCm dispatches to for that format returns wrong values (see csrc/image.hip); everything else is the module itself."""
    ks, st, pd = module.kernel_size, module.stride, module.padding
    three = ks in (3, (3, 3)) and st in (2, (2, 2)) and pd in (1, (1, 1)) and module.count_include_pad and not module.ceil_mode \
        and module.divisor_override is None
    if three and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.float16) and not x.is_contiguous() \
            and x.is_contiguous(memory_format=torch.channels_last):
        return _AvgPool3s2Rows.apply(x)
    if x.is_cuda and x.requires_grad and not x.is_contiguous():
        x = x.contiguous()                              # (any other channels-last pool: in the format whose gradient is right)
    return module(x)


class _LeakyBatchNormRows(Function):
    """BatchNorm2d_train(LeakyReLU(x)) of a channels-last map as ONE node on the library's row kernels (csrc/bn.hip,
    ts_leaky_bn_train_*): x - the convolution's output - is what is kept for the backward pass, the activated map never exists"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, eps, momentum, slope, residual=None):
        t, c, h, w = x.shape
        n = t * h * w
        half = x.dtype == torch.float16
        lib = B.L.load()
        out = torch.empty_like(x, memory_format=torch.channels_last)
        stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws = B.L.workspace(lib.ts_bn_train_workspace_bytes(c), x.device)
        B.L.check(lib.ts_leaky_bn_train_forward(B.L.ptr(x), B.L.ptr(weight), B.L.ptr(bias), B.L.ptr(running_mean), B.L.ptr(running_var),
                                                B.L.ptr(nbt), n, c, float(eps), float(momentum), float(slope), 1 if half else 0,
                                                B.L.ptr(stats[0]), B.L.ptr(stats[1]), B.L.ptr(residual), B.L.ptr(out), B.L.ptr(ws), ws.numel(),
                                                B.L.stream()), "ts_leaky_bn_train_forward")
        for buf in (running_mean, running_var, nbt):      # written through raw pointers: move their version counters
            if buf is not None:
                torch.autograd.graph.increment_version(buf)
        ctx.save_for_backward(x, weight, stats)
        ctx.slope, ctx.with_residual = float(slope), residual is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, stats = ctx.saved_tensors
        t, c, h, w = x.shape
        lib = B.L.load()
        grad_out = grad_out.to(x.dtype).contiguous(memory_format=torch.channels_last)
        grad_x = torch.empty_like(x, memory_format=torch.channels_last)
        gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws = B.L.workspace(lib.ts_bn_train_workspace_bytes(c), x.device)
        B.L.check(lib.ts_leaky_bn_train_backward(B.L.ptr(grad_out), B.L.ptr(x), B.L.ptr(stats[0]), B.L.ptr(stats[1]), B.L.ptr(weight),
                                                 t * h * w, c, ctx.slope, 1 if x.dtype == torch.float16 else 0, B.L.ptr(grad_x),
                                                 B.L.ptr(gwb[0]), B.L.ptr(gwb[1]), B.L.ptr(ws), ws.numel(), B.L.stream()),
                  "ts_leaky_bn_train_backward")
        # (the residual's gradient is the node's own incoming gradient: the sum's other branch)
        return (grad_x, gwb[0].to(weight.dtype), gwb[1].to(weight.dtype), None, None, None, None, None, None,
                grad_out if ctx.with_residual else None)


def _act_bn(act, bn, x, residual=None):
    """bn(act(x)) of a block (unet2d.py:24-30,71,108), plus the block's residual sum (`skip + y`, unet2d.py:31,64) when given.
    Training-mode BatchNorm2d behind a LeakyReLU on a channels-last device map goes through ONE fused node - the residual added in its
    elementwise pass; anything else - evaluation mode, other layouts / dtypes, modules with hooks, options.image_fused_bn off - is the
    modules themselves."""
    c = x.shape[1] if x.dim() == 4 else 0
    if (options.image_fused_bn and bn.training and x.is_cuda and x.dim() == 4 and isinstance(act, nn.LeakyReLU) and type(bn) is nn.BatchNorm2d
            and bn.affine and bn.track_running_stats and bn.momentum is not None and x.dtype in (torch.float32, torch.float16)
            and c % (8 if x.dtype == torch.float16 else 4) == 0 and c <= 1024 and bn.weight.dtype == torch.float32
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and not (act._forward_hooks or act._forward_pre_hooks or act._backward_hooks or bn._forward_hooks or bn._forward_pre_hooks
                     or bn._backward_hooks)):
        if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype
                                         and residual.is_contiguous(memory_format=torch.channels_last)):
            return residual + _act_bn(act, bn, x)
        return _LeakyBatchNormRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                                         bn.momentum, act.negative_slope, residual)
    y = bn(act(x))
    return y if residual is None else residual + y


class _Conv3x3C32Rows(Function):
    """Conv2d(32, 32, 3, stride 1, padding = dilation) of a channels-last half map on csrc/conv2d_rows.hip: forward, data gradient
    (the weight lives in a wave's registers as MFMA operands) and weight gradient (pixel-major rows through LDS transposing reads,
    partial sums in a fixed order) on the library's kernels - tools/conv2d_probe.py has them against MIOpen's best solvers for
    this shape; the bias gradient comes out of the weight gradient's pass (a product against ones)."""

    @staticmethod
    def forward(ctx, x, weight16, bias, dilation):
        y = B.conv3x3c32_rows(x, B.conv3x3c32_pack(weight16, 0), None if bias is None else bias.float(), dilation)
        ctx.save_for_backward(x, weight16)
        ctx.dilation, ctx.bias_dtype = int(dilation), None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, w = ctx.saved_tensors
        d = ctx.dilation
        grad_y = grad_y.to(torch.float16).contiguous(memory_format=torch.channels_last)
        gx = B.conv3x3c32_rows(grad_y, B.conv3x3c32_pack(w, 1), None, d) if ctx.needs_input_grad[0] else None
        gw = gb = None
        want_bias = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_bias:
            gw, gb = B.conv3x3c32_wgrad(x, grad_y, w, d, True)           # (the bias sums ride along in the same pass)
            gb = gb.to(ctx.bias_dtype) if want_bias else None
        return gx, gw if ctx.needs_input_grad[1] else None, gb, None


class _ShuffleCatRows(Function):
    """UpBlock's entry - PixelShuffle(2), Dropout2d, concat with the skip map, Dropout2d - of two channels-last maps as one pass over
    the result (csrc/shuffle_cat.hip); `scale` [T, C/4 + Cs] float32 holds the two dropout masks folded together, or None.  The
    adjoint hands back two CONTIGUOUS gradients (ATen's concat backward leaves strided slices every consumer first copies)."""

    @staticmethod
    def forward(ctx, x, skip, scale):
        ctx.channels, ctx.scale = x.shape[1], scale
        return B.shuffle_cat_rows_forward(x, skip, scale)

    @staticmethod
    def backward(ctx, grad_cat):
        grad_cat = grad_cat.contiguous(memory_format=torch.channels_last)
        gx, gs = B.shuffle_cat_rows_backward(grad_cat, ctx.channels, ctx.scale)
        return gx, gs, None


class _Conv3x3Rows(Function):
    """Conv2d(C_in <= 96, C_out, 3, stride 1, padding 1) of a channels-last half map - the decoder's wide layers (UpBlock.conv1 of up3:
    96 -> 96 at 1/2 scale, up4: 56 -> 96 at full scale) - forward and data gradient on csrc/conv2d_rows.hip's general kernel (MIOpen's
    best solvers for the up4 shape: 4.4, 1.5 and 2.4 ms, tools/unet2d_layers.py) and weight + bias gradient on its companion (fp32 sums in
    registers, partials added in a fixed order)."""

    @staticmethod
    def forward(ctx, x, weight16, bias):
        y = B.conv3x3_rows(x, B.conv3x3_rows_pack(weight16, 0), None if bias is None else bias.float(), weight16.shape[0])
        ctx.save_for_backward(x, weight16)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, w = ctx.saved_tensors
        grad_y = grad_y.to(torch.float16).contiguous(memory_format=torch.channels_last)
        gx = B.conv3x3_rows(grad_y, B.conv3x3_rows_pack(w, 1), None, w.shape[1]) if ctx.needs_input_grad[0] else None
        gw = gb = None
        want_bias = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_bias:
            gw, gb = B.conv3x3_wgrad(x, grad_y, w, want_bias)            # (the bias sums ride along in the same pass)
            gb = gb.to(ctx.bias_dtype) if want_bias else None
        return gx, gw if ctx.needs_input_grad[1] else None, gb


class _Conv1x1C32Act(Function):
    """LeakyReLU(Conv2d(32, 32, 1)) of a channels-last half map as one node (csrc/conv2d_rows.hip: the activation in the product's
    epilogue; backward: ATen's LeakyReLU gradient from the saved OUTPUT, then data gradient on the same kernel and weight + bias
    gradient on the 3 x 3 layers' pass)."""

    @staticmethod
    def forward(ctx, x, weight16, bias, slope):
        y = B.conv1x1c32_rows(x, B.conv1x1c32_pack(weight16, 0), None if bias is None else bias.float(), slope)
        ctx.save_for_backward(x, weight16, y)
        ctx.slope, ctx.bias_dtype = float(slope), None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, w, y = ctx.saved_tensors
        grad_y = grad_y.to(torch.float16).contiguous(memory_format=torch.channels_last)
        g = torch.ops.aten.leaky_relu_backward(grad_y, y, ctx.slope, True)           # (slope > 0: the output's sign is the input's)
        gx = B.conv1x1c32_rows(g, B.conv1x1c32_pack(w, 1), None, None) if ctx.needs_input_grad[0] else None
        gw = gb = None
        want_bias = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_bias:
            gw, gb = B.conv1x1c32_wgrad(x, g, w)
            gb = gb.to(ctx.bias_dtype) if want_bias else None
        return gx, gw if ctx.needs_input_grad[1] else None, gb, None


def _conv_act(conv, act, x):
    """act(conv(x)) at the head of a block: the 1 x 1, 32 -> 32 channel layers of a channels-last half map with their LeakyReLU as one
    node; everything else the two modules"""
    if (options.image_conv_rows and x.is_cuda and x.dim() == 4 and x.dtype == torch.float16 and type(conv) is nn.Conv2d
            and type(act) is nn.LeakyReLU and not act.inplace and act.negative_slope > 0
            and conv.in_channels == 32 and conv.out_channels == 32 and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.groups == 1 and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and x.shape[2] * x.shape[3] >= _CONV_ROWS_MIN_PIXELS
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and not (conv._forward_hooks or conv._forward_pre_hooks or conv._backward_hooks or act._forward_hooks or act._forward_pre_hooks)):
        return _Conv1x1C32Act.apply(x, conv.weight.to(torch.float16), conv.bias, act.negative_slope)
    return act(conv(x))


# layers the general kernel is measured faster on (tools/conv2d_probe.py): at least this many pixels per call
_CONV_ROWS_MIN_PIXELS = 192 * 640


def _conv(conv, x):
    """the block's Conv2d.  The 3 x 3, 32 -> 32 channel layers (plain and dilated: the seven full-resolution layers of the stem and
    stage 1) of a channels-last half map - what autocast makes of them - go through the library's kernel; everything else is the
    module itself."""
    if (options.image_conv_rows and x.is_cuda and x.dim() == 4 and x.dtype == torch.float16 and type(conv) is nn.Conv2d
            and conv.in_channels == 32 and conv.out_channels == 32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.groups == 1 and conv.padding == conv.dilation and conv.dilation in ((1, 1), (2, 2)) and conv.padding_mode == "zeros"
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and not (conv._forward_hooks or conv._forward_pre_hooks or conv._backward_hooks)):
        return _Conv3x3C32Rows.apply(x, conv.weight.to(torch.float16), conv.bias, conv.dilation[0])
    if (options.image_conv_rows and x.is_cuda and x.dim() == 4 and x.dtype == torch.float16 and type(conv) is nn.Conv2d
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.groups == 1 and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.padding_mode == "zeros" and x.shape[2] * x.shape[3] >= _CONV_ROWS_MIN_PIXELS
            and (conv.in_channels, conv.out_channels) != (32, 32) and B.conv3x3_rows_takes(conv.in_channels, conv.out_channels)
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and not (conv._forward_hooks or conv._forward_pre_hooks or conv._backward_hooks)):
        return _Conv3x3Rows.apply(x, conv.weight.to(torch.float16), conv.bias)
    return conv(x)


def _leaky():
    return nn.LeakyReLU()


class ResContextBlock(nn.Module):
    """1x1 -> (3x3 -> BN) -> (3x3 dilated 2 -> BN), residual on the 1x1 branch (unet2d.py:7-31)."""

    def __init__(self, in_filters, out_filters):
        super().__init__()
        self.conv1 = nn.Conv2d(in_filters, out_filters, kernel_size=(1, 1), stride=1)
        self.act1 = _leaky()
        self.conv2 = nn.Conv2d(out_filters, out_filters, (3, 3), padding=1)
        self.act2 = _leaky()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, (3, 3), dilation=2, padding=2)
        self.act3 = _leaky()
        self.bn2 = nn.BatchNorm2d(out_filters)

    def forward(self, x):
        skip = _conv_act(self.conv1, self.act1, x)
        y = _act_bn(self.act2, self.bn1, _conv(self.conv2, skip))
        return _act_bn(self.act3, self.bn2, _conv(self.conv3, y), residual=skip)


class ResBlock(nn.Module):
    """1x1 shortcut + (3x3 -> BN), optional Dropout2d and 3x3 / stride-2 average pool (unet2d.py:34-78)."""

    def __init__(self, in_filters, out_filters, dropout_rate, kernel_size=(3, 3), stride=1, pooling=True, drop_out=True,
                 return_skip=True):
        super().__init__()
        self.pooling, self.drop_out, self.return_skip = pooling, drop_out, return_skip
        self.conv1 = nn.Conv2d(in_filters, out_filters, kernel_size=(1, 1), stride=stride)
        self.act1 = _leaky()
        self.conv2 = nn.Conv2d(in_filters, out_filters, kernel_size=(3, 3), padding=1)
        self.act2 = _leaky()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.dropout = nn.Dropout2d(p=dropout_rate)
        if pooling:
            self.pool = nn.AvgPool2d(kernel_size=kernel_size, stride=2, padding=1)

    def forward(self, x):
        res = _act_bn(self.act2, self.bn1, _conv(self.conv2, x), residual=_conv_act(self.conv1, self.act1, x))
        out = self.dropout(res) if self.drop_out else res
        if not self.pooling:
            return out
        out = _pool(self.pool, out)
        return (out, res) if self.return_skip else out


class UpBlock(nn.Module):
    """PixelShuffle(2) + skip concat + (3x3 -> BN) with Dropout2d around it (unet2d.py:81-115)."""

    def __init__(self, in_filters, out_filters, dropout_rate=0.2, drop_out=True, mid_filters=None):
        super().__init__()
        self.drop_out, self.in_filters, self.out_filters = drop_out, in_filters, out_filters
        self.mid_filters = mid_filters if mid_filters else in_filters // 4 + 2 * out_filters
        self.dropout1 = nn.Dropout2d(p=dropout_rate)
        self.dropout2 = nn.Dropout2d(p=dropout_rate)
        self.conv1 = nn.Conv2d(self.mid_filters, out_filters, (3, 3), padding=1)
        self.act1 = _leaky()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.dropout3 = nn.Dropout2d(p=dropout_rate)
        self.shuffle = nn.PixelShuffle(2)       # parameter-free (the reference builds it inside forward)

    def _masks(self, x, skip):
        """the two Dropout2d masks of forward() as one float32 factor [T, C/4 + Cs] (same distribution: a Bernoulli draw per frame and
        channel of `upA` and of the concatenation, kept values scaled by 1 / (1 - p)), or None when nothing is dropped"""
        if not (self.drop_out and (self.dropout1.training or self.dropout2.training)):      # (the modules' own flags, as in their forward)
            return None
        t, cq, cs = x.shape[0], x.shape[1] // 4, skip.shape[1]
        ones = torch.ones((t, cq + cs, 1, 1), dtype=torch.float32, device=x.device)
        m1 = nn.functional.dropout2d(ones[:, :cq], self.dropout1.p, self.dropout1.training)
        m2 = nn.functional.dropout2d(ones, self.dropout2.p, self.dropout2.training)
        return (torch.cat((m1, ones[:, cq:]), dim=1) * m2).reshape(t, cq + cs)

    def forward(self, x, skip):
        if options.image_shuffle_cat and B.shuffle_cat_rows_takes(x, skip):
            cat = _ShuffleCatRows.apply(x, skip, self._masks(x, skip))
        else:
            up = self.shuffle(x)
            if self.drop_out:
                up = self.dropout1(up)
            cat = torch.cat((up, skip), dim=1)
            if self.drop_out:
                cat = self.dropout2(cat)
        out = _act_bn(self.act1, self.bn1, _conv(self.conv1, cat))
        return self.dropout3(out) if self.drop_out else out


class UNet2D(nn.Module):
    """unet2d.py:118-216.  `forward(data_dict)` adds `image_logits` and the per-FOV-point gathers
    `image_logits_fov`, `image_targets_fov`, `image_rgb_fov`, `image_features_fov` (= 96 full-scale channels of the
    last decoder stage + 128 channels of the 1/4-scale stage)."""

    def __init__(self, input_dim=3, num_class=20):
        super().__init__()
        self.input_dim, self.num_class = input_dim, num_class
        self.cr = 1.0
        self.cs = cs = [int(self.cr * x) for x in [32, 32, 64, 128, 256, 256, 128, 96, 96]]
        self.stem = nn.Sequential(ResContextBlock(input_dim, cs[0]), ResContextBlock(cs[0], cs[0]),
                                  ResContextBlock(cs[0], cs[0]))
        self.stage1 = ResBlock(cs[0], cs[1], 0.2, pooling=True, drop_out=False)
        self.stage2 = ResBlock(cs[1], cs[2], 0.2, pooling=True)
        self.stage3 = ResBlock(cs[2], cs[3], 0.2, pooling=True)
        self.stage4 = ResBlock(cs[3], cs[4], 0.2, pooling=True)
        self.mid_stage = ResBlock(cs[4], cs[4], 0.2, pooling=False)
        self.up1 = UpBlock(cs[4], cs[5], 0.2, mid_filters=cs[4] // 4 + cs[4])
        self.up2 = UpBlock(cs[5], cs[6], 0.2, mid_filters=cs[5] // 4 + cs[3])
        self.up3 = UpBlock(cs[6], cs[7], 0.2, mid_filters=cs[6] // 4 + cs[2])
        self.up4 = UpBlock(cs[7], cs[8], 0.2, drop_out=False, mid_filters=cs[7] // 4 + cs[1])
        self.classifier = nn.Sequential(nn.Conv2d(cs[8], num_class, kernel_size=1, stride=1))
        self._layout = None

    def _set_layout(self):
        """parameters in the memory format of options.image_layout (once; the Parameter objects stay)"""
        want = torch.channels_last if options.image_layout == "nhwc" else torch.contiguous_format
        if self._layout is not want:
            self.to(memory_format=want)
            self._layout = want
        return want

    # the dense network in three pieces around the two maps the hand-over reads (unet2d.py:155-172); a test can stand in for them
    def _encode(self, x):
        x0 = self.stem(x)
        x1, s1 = self.stage1(x0)
        x2, s2 = self.stage2(x1)
        x3, s3 = self.stage3(x2)
        x4, s4 = self.stage4(x3)
        return self.mid_stage(x4), (s1, s2, s3, s4)

    def _decode_u2(self, x5, skips):
        return self.up2(self.up1(x5, skips[3]), skips[2])           # 1/4 scale, 128 channels

    def _decode_u4(self, u2, skips):
        return self.up4(self.up3(u2, skips[1]), skips[0])           # full scale, 96 channels

    def forward(self, data_dict):
        x = data_dict["image_input"]
        x = x.contiguous(memory_format=self._set_layout())
        height, width = int(x.shape[2]), int(x.shape[3])
        x5, skips = self._encode(x)
        fov = data_dict["lidar_fov_ms"]
        pix = fov.F[:, -2:].float().contiguous()                      # (row in the sample's stacked frames, col)
        pbatch = fov.C[:, -1].int().contiguous()
        frame_end = torch.as_tensor(data_dict["offset_img"], device=x.device).int().contiguous()
        with torch.no_grad():       # the points in raster order, once per scale (csrc/image.hip)
            plan0 = image_plan(pix, pbatch, frame_end, x.shape[0], height, width, 0)
            plan4 = image_plan(pix, pbatch, frame_end, x.shape[0], height, width, 2)
        u2 = self._decode_u2(x5, skips)
        # (the gathered maps go on THROUGH their gather nodes: the adjoint is added into the gradient they get from here on)
        u2, feat4 = _image_gather_through(u2, plan4)
        u4 = self._decode_u4(u2, skips)
        u4, feat0 = _image_gather_through(u4, plan0)
        logits = self.classifier(u4)
        logits, logits_fov = _image_gather_through(logits, plan0)
        data_dict["image_logits"] = logits
        err = plan0["err"]
        with torch.no_grad():
            targets_fov, _ = image_gather(data_dict["semantic_map_ms"], None, None, None, height, width, plan=plan0)
            rgb_fov, _ = image_gather(x, None, None, None, height, width, plan=plan0)
        data_dict["image_logits_fov"] = logits_fov
        data_dict["image_targets_fov"] = targets_fov[:, 0].to(data_dict["semantic_map_ms"].dtype)
        data_dict["image_rgb_fov"] = rgb_fov
        data_dict["image_features_fov"] = torch.cat([feat0, feat4], dim=-1)
        data_dict["image_gather_err"] = err     # non-zero: a point projects outside its sample's frames
        return data_dict
