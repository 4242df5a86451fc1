"""UNet3D - the FOV-cloud encoder of TIAF (reference pcseg/model/segmentor/voxel/minkunet/unet3d.py:180-316).

A sparse encoder (stem + 4 strided stages of one residual block each, 96-96-128-128-256 channels) over the points
inside the camera frustum, whose input features are the LiDAR attributes concatenated with the gathered image
features.  It returns its own point logits and the stride-16 / stride-4 / stride-1 voxel features that
`MinkUNetMsMm` devoxelises onto the full cloud (`voxel_to_point_fov`).  Same module tree / state_dict as the
reference; every sparse op runs on the HIP backend.
"""
import torch
from torch import nn

from taseg_amd.torchsparse import PointTensor, SparseTensor
from taseg_amd.torchsparse import nn as spnn
from taseg_amd.torchsparse.nn import functional as spF
from .minkunet import BasicConvolutionBlock, ResidualBlock, _norm
from .utils import voxel_to_point

__all__ = ["UNet3D"]


class UNet3D(nn.Module):
    def __init__(self, input_dim=96, num_class=20, if_dist=True):
        super().__init__()
        self.in_feature_dim, self.num_class = input_dim, num_class
        self.num_layer = [1] * 8
        self.block = ResidualBlock
        cs = [96, 96, 128, 128, 256, 256, 128, 96, 96]
        self.pres = self.vres = 0.05
        self.stem = nn.Sequential(
            spnn.Conv3d(input_dim, cs[0], kernel_size=3, stride=1), _norm(cs[0], if_dist), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), _norm(cs[0], if_dist), spnn.ReLU(True))
        self.in_channels = cs[0]

        def stage(width, depth):
            down = BasicConvolutionBlock(self.in_channels, self.in_channels, ks=2, stride=2, dilation=1, if_dist=if_dist)
            blocks = [self.block(self.in_channels, width, if_dist=if_dist)]
            self.in_channels = width * self.block.expansion
            blocks += [self.block(self.in_channels, width, if_dist=if_dist) for _ in range(1, depth)]
            return nn.Sequential(down, *blocks)

        self.stage1 = stage(cs[1], self.num_layer[0])
        self.stage2 = stage(cs[2], self.num_layer[1])
        self.stage3 = stage(cs[3], self.num_layer[2])
        self.stage4 = stage(cs[4], self.num_layer[3])
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.dropout = nn.Dropout(0.0, True)
        self.classifier = nn.Sequential(spnn.PointLinear((cs[0] + cs[2] + cs[4]) * self.block.expansion, num_class))

    def forward(self, batch_dict):
        x = batch_dict["lidar_fov_ms"]
        x.F = x.F[:, :self.in_feature_dim]
        z = PointTensor(x.F, x.C.float())
        x = SparseTensor(x.F, x.C, x.s)                       # fresh caches: the FOV cloud has its own rulebooks
        spF.build_pyramid(x, num_levels=4)
        x0 = spnn.conv_bn_act(self.stem[0], self.stem[1], x, relu=True)
        x0 = spnn.conv_bn_act(self.stem[3], self.stem[4], x0, relu=True)
        z0 = voxel_to_point(x0, z, nearest=False)
        x1 = self.stage1(x0)
        x2 = self.stage2(x1)
        z1 = voxel_to_point(x2, z0)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        z2 = voxel_to_point(x4, z1)
        out = self.classifier(torch.cat([z0.F, z1.F, z2.F], dim=1))
        return out, x4, x2, x0
