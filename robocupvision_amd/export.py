"""Robot-side export of a trained network: the flat float64 weight file of paramSave.py and the
darknet-style layer list of weights/net.cfg that the on-robot inference engine reads next to it.  Host-side I/O, no kernels."""
from __future__ import annotations

import os

import numpy as np
import torch

from .model import PB_FCN, ConvPool, ConvPoolSimple, upSampleTransposeConv


def flat_params(model, skipClassifier: bool = False) -> np.ndarray:
    """Every state_dict entry (parameters AND buffers, num_batches_tracked included) flattened and concatenated in
    state_dict order as float64 (paramSave.py:9-18); entries whose name contains 'classifier' are dropped on request."""
    chunks = []
    for name, v in model.state_dict().items():
        if "classifier" in name and skipClassifier:
            continue
        chunks.append(v.detach().cpu().numpy().reshape(-1).astype(np.float64))
    return np.concatenate(chunks) if chunks else np.empty(0)


def saveParams(path, model, fName="weights.dat", skipClassifier=False):
    """paramSave.py:5-18: raw float64 dump to path/fName."""
    if not os.path.exists(path):
        os.makedirs(path)
    flat_params(model, skipClassifier).tofile(os.path.join(path, fName))


def _conv_cfg(conv, activation, has_bias=None):
    lines = ["[convolutional]", "filters=%d" % conv.out_channels, "size=%d" % conv.kernel_size[0], "stride=%d" % conv.stride[0],
             "pad=%d" % conv.padding[0]]
    if has_bias is not None:
        lines.append("dilation=%d" % conv.dilation[0])
    lines.append("activation=%s" % activation)
    if has_bias is not None:
        lines.append("hasBias=%d" % has_bias)
    return lines


def net_cfg(model: PB_FCN) -> str:
    """The layer list of weights/net.cfg for a PB_FCN: one [convolutional]/[transposedconv] section per conv, a [batchnorm]
    section where one follows, [shortcut] sections for the skip adds (``from`` = index of the encoder block), [softmax] last."""
    if not isinstance(model, PB_FCN):
        raise TypeError("net_cfg describes PB_FCN networks")
    H, W = model.img_shape
    sec = [["[net]", "height=%d" % H, "width=%d" % W, "channels=3", "downscale=4"]]
    bn = ["[batchnorm]", "activation = relu"]
    enc_index = {}          # encoder block name -> layer index its [shortcut] refers to
    enc = model.FCN
    order = ["conv0", "conv1", "conv2"] + (["conv_ext"] if model.noScale else []) + ["conv3", "conv4", "conv5", "conv6", "conv7", "conv8"]
    idx = 0
    for name in order:
        m = getattr(enc, name)
        if isinstance(m, ConvPoolSimple):
            sec.append(_conv_cfg(m.conv, "linear", 0 if m.conv.bias is None else 1))
            sec.append(bn)
            idx += 2
        elif isinstance(m, ConvPool):
            sec.append(_conv_cfg(m.conv1, "relu", 0))
            sec.append(_conv_cfg(m.pool, "linear", 0))
            sec.append(bn)
            idx += 3
        enc_index[name] = idx - 1       # layer index (0-based, [net] not counted) of the block's closing [batchnorm]
    ups = [model.up1, model.up2, model.up3] + ([model.up4] if model.noScale else [])
    skips = (["conv_ext", "conv2", "conv1", "conv0"] if model.noScale else ["conv2", "conv1", "conv0"])
    for up, skip in zip(ups, skips):
        assert isinstance(up, upSampleTransposeConv)
        c = up.conv
        sec.append(["[transposedconv]", "filters=%d" % c.out_channels, "size=%d" % c.kernel_size[0], "stride=%d" % c.stride[0],
                    "pad=%d" % c.padding[0], "outpad=%d" % c.output_padding[0], "activation=linear"])
        sec.append(bn)
        sec.append(["[shortcut]", "activation=linear", "from=%d" % enc_index[skip]])
    sec.append(_conv_cfg(model.segmenter.classifier, "linear"))
    sec.append(["[softmax]"])
    return "\n\n".join("\n".join(lines) for lines in sec) + "\n"
