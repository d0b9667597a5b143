"""FPN necks.  ``FPN2`` mirrors mmdet/models/necks/fpn2.py:10-141 (an FPN that returns only the
pyramid levels listed in ``select_out``); ``FPN`` is the same module returning every level.

All parameters of the reference exist under the same names (``lateral_convs.{i}.conv|gn``,
``fpn_convs.{i}.conv|gn``) so checkpoints load, but branches that cannot reach a selected output
are not evaluated: with KGDet's ``select_out=[2]`` only ``lateral_convs[2]`` and ``fpn_convs[2]``
run, which is ~34 GFLOP/image of dead work the reference executes.  Outputs are
bit-identical because the top-down additions only flow from coarse to fine levels.  The skipped
modules' parameters receive no gradient, exactly as in the reference.
"""
import torch.nn as nn
import torch.nn.functional as F

from .layers import ConvModule, xavier_init
from .registry import NECKS


@NECKS.register_module
class FPN2(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 select_out=[0, 1, 2, 3, 4], add_extra_convs=False, extra_convs_on_inputs=True,
                 relu_before_extra_convs=False, conv_cfg=None, norm_cfg=None, activation=None):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.select_out = select_out
        self.activation = activation
        self.relu_before_extra_convs = relu_before_extra_convs
        self.fp16_enabled = False

        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level = start_level
        self.end_level = end_level
        self.add_extra_convs = add_extra_convs
        self.extra_convs_on_inputs = extra_convs_on_inputs

        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, activation=self.activation, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, activation=self.activation, inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                if i == 0 and self.extra_convs_on_inputs:
                    chn = self.in_channels[self.backbone_end_level - 1]
                else:
                    chn = out_channels
                self.fpn_convs.append(ConvModule(chn, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, activation=self.activation, inplace=False))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def _output(self, idx, inputs, cache):
        """pyramid output `idx`, computing only what it depends on (memoised in `cache`)"""
        if ('out', idx) in cache:
            return cache[('out', idx)]
        n_lat = len(self.lateral_convs)

        def lateral(i):  # top-down merged lateral i depends on laterals i .. n_lat-1
            if ('lat', i) not in cache:
                x = self.lateral_convs[i](inputs[i + self.start_level])
                if i + 1 < n_lat:
                    x = x + F.interpolate(lateral(i + 1), scale_factor=2, mode='nearest')
                cache[('lat', i)] = x
            return cache[('lat', i)]

        if idx < n_lat:
            out = self.fpn_convs[idx](lateral(idx))
        elif not self.add_extra_convs:
            out = F.max_pool2d(self._output(idx - 1, inputs, cache), 1, stride=2)
        elif idx == n_lat:
            src = inputs[self.backbone_end_level - 1] if self.extra_convs_on_inputs else \
                self._output(idx - 1, inputs, cache)
            out = self.fpn_convs[idx](src)
        else:
            prev = self._output(idx - 1, inputs, cache)
            out = self.fpn_convs[idx](F.relu(prev) if self.relu_before_extra_convs else prev)
        cache[('out', idx)] = out
        return out

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        cache = {}
        return tuple(self._output(idx, inputs, cache) for idx in self.select_out)


@NECKS.register_module
class FPN(FPN2):
    """mmdet/models/necks/fpn.py: every level is returned"""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, conv_cfg=None, norm_cfg=None,
                 activation=None):
        super().__init__(in_channels, out_channels, num_outs, start_level, end_level, list(range(num_outs)),
                         add_extra_convs, extra_convs_on_inputs, relu_before_extra_convs, conv_cfg, norm_cfg,
                         activation)
