"""UNet2D - the camera branch of TIAF (reference pcseg/model/segmentor/voxel/minkunet/unet2d.py).

The dense 2-D convolutions / BatchNorm2d / pooling / PixelShuffle stay on PyTorch-ROCm (MIOpen / rocBLAS are the
roofline for dense NCHW work - SURVEY.md section 2.1 row 4); module tree and parameter names are the reference's, so
its checkpoints load (`stem.0.conv1.weight`, `stage2.bn1.running_mean`, `up3.conv1.weight`, `classifier.0.bias`).

What is rebuilt is the image -> point hand-over (unet2d.py:180-214): the reference materialises NHWC copies of five
feature stacks (6.8 GB for the 96-channel full-resolution map of 36 frames) and fancy-indexes them per sample;
`image_gather` reads the NCHW stacks in place with one HIP kernel per map (`ts_image_gather_forward`) and
accumulates its adjoint with atomics (`ts_image_gather_backward`).
"""
import torch
from torch import nn
from torch.autograd import Function

from taseg_amd import backend as B

__all__ = ["UNet2D", "image_gather"]


class _ImageGather(Function):
    @staticmethod
    def forward(ctx, feat, pix, pbatch, frame_end, height, width, shift):
        out, err = B.image_gather_forward(feat, pix, pbatch, frame_end, height, width, shift)
        ctx.saved = (pix, pbatch, frame_end, feat.shape[0], height, width, shift)
        ctx.mark_non_differentiable(err)
        return out, err

    @staticmethod
    def backward(ctx, grad_out, _grad_err):
        pix, pbatch, frame_end, frames, height, width, shift = ctx.saved
        grad = B.image_gather_backward(grad_out.contiguous(), pix, pbatch, frame_end, frames, height, width, shift)
        return grad, None, None, None, None, None, None


def image_gather(feat, pix, pbatch, frame_end, height, width, shift=0):
    """Rows of `feat` [T, C, H >> shift, W >> shift] at the pixels the FOV points project to -> ([n, C], err)."""
    return _ImageGather.apply(feat.contiguous().float(), pix, pbatch, frame_end, int(height), int(width), int(shift))


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
        skip = self.act1(self.conv1(x))
        y = self.bn1(self.act2(self.conv2(skip)))
        y = self.bn2(self.act3(self.conv3(y)))
        return skip + y


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
        res = self.act1(self.conv1(x)) + self.bn1(self.act2(self.conv2(x)))
        out = self.dropout(res) if self.drop_out else res
        if not self.pooling:
            return out
        out = self.pool(out)
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

    def forward(self, x, skip):
        up = self.shuffle(x)
        if self.drop_out:
            up = self.dropout1(up)
        cat = torch.cat((up, skip), dim=1)
        if self.drop_out:
            cat = self.dropout2(cat)
        out = self.bn1(self.act1(self.conv1(cat)))
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

    def forward(self, data_dict):
        x = data_dict["image_input"]
        height, width = int(x.shape[2]), int(x.shape[3])
        x0 = self.stem(x)
        x1, s1 = self.stage1(x0)
        x2, s2 = self.stage2(x1)
        x3, s3 = self.stage3(x2)
        x4, s4 = self.stage4(x3)
        x5 = self.mid_stage(x4)
        u1 = self.up1(x5, s4)
        u2 = self.up2(u1, s3)           # 1/4 scale, 128 channels
        u3 = self.up3(u2, s2)
        u4 = self.up4(u3, s1)           # full scale, 96 channels
        logits = self.classifier(u4)
        data_dict["image_logits"] = logits

        fov = data_dict["lidar_fov_ms"]
        pix = fov.F[:, -2:].float().contiguous()                      # (row in the sample's stacked frames, col)
        pbatch = fov.C[:, -1].int().contiguous()
        frame_end = torch.as_tensor(data_dict["offset_img"], device=x.device).int().contiguous()
        args = (pix, pbatch, frame_end, height, width)
        logits_fov, err = image_gather(logits, *args)
        with torch.no_grad():
            targets_fov, _ = image_gather(data_dict["semantic_map_ms"].float(), *args)
            rgb_fov, _ = image_gather(x, *args)
        feat0, _ = image_gather(u4, *args)
        feat4, _ = image_gather(u2, *args, shift=2)
        data_dict["image_logits_fov"] = logits_fov
        data_dict["image_targets_fov"] = targets_fov[:, 0].to(data_dict["semantic_map_ms"].dtype)
        data_dict["image_rgb_fov"] = rgb_fov
        data_dict["image_features_fov"] = torch.cat([feat0, feat4], dim=-1)
        data_dict["image_gather_err"] = err     # non-zero: a point projects outside its sample's frames
        return data_dict
