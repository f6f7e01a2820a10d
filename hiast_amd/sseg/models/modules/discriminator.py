"""FCDiscriminator (reference: sseg/models/modules/discriminator.py:7-33): five 4x4 stride-2 convolutions,
C -> 64 -> 128 -> 256 -> 512 -> 1, LeakyReLU(0.2) between them; state-dict names conv1..conv4, classifier.
31 GFLOP per 512x1024 image (4 % of one trunk forward): the convolutions go through MIOpen via PyTorch-ROCm."""
from torch import nn
from torch.nn import functional as F

__all__ = ["build_discriminator", "FCDiscriminator"]


class FCDiscriminator(nn.Module):

    def __init__(self, num_classes, ndf=64):
        super().__init__()
        chans = [num_classes, ndf, ndf * 2, ndf * 4, ndf * 8]
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(chans[i], chans[i + 1], kernel_size=4, stride=2, padding=1))
        self.classifier = nn.Conv2d(ndf * 8, 1, kernel_size=4, stride=2, padding=1)
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2, inplace=True)

    def forward(self, x, params=None):
        """params: optional {name: tensor} overriding the module's own (used with detached weights when only the
        gradient w.r.t. the input is wanted)"""
        for name in ("conv1", "conv2", "conv3", "conv4", "classifier"):
            m = getattr(self, name)
            wgt = m.weight if params is None else params[name + ".weight"]
            b = m.bias if params is None else params[name + ".bias"]
            x = F.conv2d(x, wgt, b, stride=2, padding=1)
            if name != "classifier":
                x = F.leaky_relu(x, 0.2, inplace=True)
        return x


def build_discriminator(input_channels):
    return FCDiscriminator(input_channels)
