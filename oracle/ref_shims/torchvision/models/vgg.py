"""Shim of torchvision.models.vgg.vgg19 (cfg 'E', no BN).  ``features`` has the
public layer layout (indices 0..36); the classifier is shrunk -- the reference
never evaluates it (model/VGG.py:13-28 only slices ``features``)."""
from torch import nn

_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_E:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.classifier = nn.Sequential(nn.Linear(512, 8))

    def forward(self, x):
        return self.classifier(self.avgpool(self.features(x)).flatten(1))


def vgg19(pretrained=False, **kwargs):
    return VGG()
