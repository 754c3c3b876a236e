from collections import OrderedDict

from torch import nn


class IntermediateLayerGetter(nn.ModuleDict):
    """Runs the children of `model` in registration order, returning the outputs named in return_layers;
    children after the last requested one are dropped (avgpool / fc never run)."""

    def __init__(self, model, return_layers):
        remaining = dict(return_layers)
        kept = OrderedDict()
        for name, module in model.named_children():
            kept[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(kept)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
