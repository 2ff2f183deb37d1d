"""``BaseNetwork.init_weights`` with the reference's traversal semantics
(SGH/model/pointnets/networks_base.py:13-63; SGH =
scene_graph_prediction/scene_graph_helpers): the initialiser visits children with
``Module.apply`` (post-order), then the network itself, skips modules whose class
name does not contain ``target_op`` and modules already marked ``param_inited``."""
import torch.nn as nn

_INITS = {
    "normal": lambda w, gain: nn.init.normal_(w, 0.0, gain),
    "xavier_normal": lambda w, gain: nn.init.xavier_normal_(w, gain=gain),
    "kaiming": lambda w, gain: nn.init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w, gain: nn.init.orthogonal_(w, gain=gain),
    "xavier_unifrom": lambda w, gain: nn.init.xavier_uniform_(w, gain=gain),   # (sic) reference spelling
    "constant": lambda w, gain: nn.init.constant_(w, gain),
}


class BaseNetwork(nn.Module):
    def init_weights(self, init_type="normal", gain=0.02, bias_value=0.0, target_op=None):
        if init_type not in _INITS:
            raise NotImplementedError(init_type)

        def visit(m):
            if target_op is not None and target_op not in m.__class__.__name__:
                return
            if hasattr(m, "param_inited"):
                return
            if hasattr(m, "weight"):
                _INITS[init_type](m.weight.data, gain)
            if hasattr(m, "bias") and m.bias is not None:
                nn.init.constant_(m.bias.data, bias_value)
            m.param_inited = True

        self.init_apply(visit)

    def init_apply(self, fn):
        for child in self.children():
            if hasattr(child, "param_inited"):
                if child.param_inited is False:
                    child.init_apply(fn)
            else:
                child.apply(fn)
        fn(self)
        return self

    def getParamList(self, x):
        return list(x.parameters())
