"""ResNet v1.5 bottleneck networks with torchvision's attribute names (conv1, bn1, layer1..4,
layerN.M.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}), stride on the 3x3 conv."""
from torch import nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, blocks, replace_stride_with_dilation=None, norm_layer=None, num_classes=1000):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        rswd = replace_stride_with_dilation or [False, False, False]
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, blocks[0], 1, False, norm_layer)
        self.layer2 = self._stage(128, blocks[1], 2, rswd[0], norm_layer)
        self.layer3 = self._stage(256, blocks[2], 2, rswd[1], norm_layer)
        self.layer4 = self._stage(512, blocks[3], 2, rswd[2], norm_layer)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, num_classes)

    def _stage(self, planes, n, stride, dilate, norm_layer):
        prev_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               norm_layer(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, prev_dilation, ds, norm_layer)]
        self.inplanes = planes * 4
        for _ in range(1, n):
            layers.append(Bottleneck(self.inplanes, planes, 1, self.dilation, None, norm_layer))
        return nn.Sequential(*layers)


def resnet50(pretrained=False, **kw):
    return ResNet((3, 4, 6, 3), **kw)


def resnet101(pretrained=False, **kw):
    return ResNet((3, 4, 23, 3), **kw)
