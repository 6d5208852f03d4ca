"""TEST / BENCH INFRASTRUCTURE ONLY: the IST backbone restated in plain PyTorch (the checker of the HIP convolutions, and what
bench.py's `cpu_baseline` port times).  Follows the reference forward line by line -- `ResNet.forward`
(/root/reference/src/models/network/resnet.py:364-381) and `BasicBlock.forward` (resnet.py:26-50) -- on the parameters of a
`gigapose_amd.ist_net.ResNet` mirror (same state-dict names as the reference class), in whatever dtype the module holds (float32 for
the parity tests, `.double()` for the float64 yardsticks).  The product package holds NO torch arithmetic for this network:
`gigapose_amd.ist_net.ResNet.forward` is HIP only and raises on CPU tensors (round 5: this file used to be a method of the product
class, `ResNet.reference_forward`)."""
import torch.nn.functional as F


def basic_block(blk, x):
    """resnet.py:40-50: conv3x3 - bn - relu - conv3x3 - bn, (+ 1x1 stride-2 downsample of the input when the block strides), add, relu."""
    y = blk.bn2(blk.conv2(F.relu(blk.bn1(blk.conv1(x)))))
    if blk.downsample is not None:
        x = blk.downsample(x)
    return F.relu(x + y)


def resnet_forward(backbone, x):
    """resnet.py:364-381: bilinear resize to input_size (align_corners=True), conv7x7/2 - bn - relu, four stages of two BasicBlocks,
    1x1 output convolution -> (b, 256, 16, 16).  BatchNorm in eval mode (the module must be in .eval())."""
    x = F.interpolate(x, (backbone.input_size, backbone.input_size), mode="bilinear", align_corners=True)
    x = F.relu(backbone.bn1(backbone.conv1(x)))
    for stage in (backbone.layer1, backbone.layer2, backbone.layer3, backbone.layer4):
        for blk in stage:
            x = basic_block(blk, x)
    return backbone.layer4_outconv(x)
