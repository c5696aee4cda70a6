"""Generates tests/golden/onnx/*: REAL exports (torch.onnx.export, the TorchScript exporter every `pyinfinitensor` example
feeds OnnxStub with) of the layer types of BASELINE configs 3 / 4, with inputs and torch's own fp32 outputs, so that the
front-end FORM the launch planner has to recognise is pinned to what an exporter really emits instead of to a reading of
pyinfinitensor/onnx.py (round-3 verdict, "do this" #7).

Run here (the build container): python tests/golden/make_onnx_fixtures.py
  * HF `BertLayer` (transformers, eager attention), small widths (hidden 128, 2 heads of 64 — the head size of BERT-base —, FFN 256) — the node sequence does not
    depend on the widths — at opset 13 (LayerNorm and Gelu decomposed into primitives) and opset 17 (LayerNormalization).
  * a ResNet stem + two torchvision-style Bottlenecks (one with a strided down-sampling branch) + global pool + classifier,
    eval mode: the exporter folds BatchNorm into the Convs, which then carry a bias — onnx.py:159-190 lowers that to
    conv -> reshape(bias) -> add.
The image has no `onnx` package; the exporter only needs it to splice onnx-script functions into the finished proto
(`_add_onnxscript_fn`), which these models do not use: that step is replaced by the identity. The bytes are what
`torch.onnx.export` serialised. No Llama fixture: the reference's front-end expects its own fused RMSNorm / RoPE /
AttentionKVCache nodes (onnx.py:368-373, 781-800), which only its out-of-tree conversion scripts emit."""
import io
from pathlib import Path

import numpy as np
import torch
from torch import nn

OUT = Path(__file__).resolve().parent / "onnx"


def export(model, args, names, opset):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes  # no `onnx` package here
    f = io.BytesIO()
    torch.onnx.export(model, args, f, dynamo=False, opset_version=opset, do_constant_folding=True, input_names=names,
                      output_names=["y"])
    return f.getvalue()


class Bottleneck(nn.Module):  # torchvision.models.resnet.Bottleneck, written out (no torchvision in the image)
    def __init__(self, cin, width, stride, down):
        super().__init__()
        cout = width * 4
        self.conv1, self.bn1 = nn.Conv2d(cin, width, 1, bias=False), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = nn.Conv2d(width, width, 3, stride, 1, bias=False), nn.BatchNorm2d(width)
        self.conv3, self.bn3 = nn.Conv2d(width, cout, 1, bias=False), nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout)) if down else None

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class TinyResNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, 16, 7, 2, 3, bias=False), nn.BatchNorm2d(16)
        self.relu, self.maxpool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
        self.layer = nn.Sequential(Bottleneck(16, 8, 1, True), Bottleneck(32, 8, 1, False), Bottleneck(32, 16, 2, True))
        self.avgpool, self.fc = nn.AdaptiveAvgPool2d((1, 1)), nn.Linear(64, 10)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.avgpool(self.layer(x))
        return self.fc(torch.flatten(x, 1))


def main():
    OUT.mkdir(exist_ok=True)
    torch.manual_seed(0)
    from transformers.models.bert.modeling_bert import BertConfig, BertLayer

    cfg = BertConfig(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=1, attn_implementation="eager")
    m = BertLayer(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():  # non-trivial LayerNorm parameters (the exporter would otherwise share the two identical ones)
            p.copy_(torch.randn_like(p) * (0.2 if p.dim() > 1 else 0.5) + (1.0 if "LayerNorm.weight" in "" else 0.0))
        m.attention.output.LayerNorm.weight.add_(1.0)
        m.output.LayerNorm.weight.add_(1.0)
    x = torch.randn(2, 16, 128)
    mask = torch.zeros(2, 1, 1, 16)
    mask[1, 0, 0, -5:] = -10000.0
    with torch.no_grad():
        y = m(x, mask)
        y = y[0] if isinstance(y, tuple) else y
    for opset in (13, 17):
        (OUT / f"bert_layer_tiny_opset{opset}.onnx").write_bytes(export(m, (x, mask), ["x", "mask"], opset))
    np.savez(OUT / "bert_layer_tiny_io.npz", x=x.numpy(), mask=mask.numpy(), y=y.numpy())

    torch.manual_seed(1)
    r = TinyResNet().eval()
    with torch.no_grad():
        for mod in r.modules():
            if isinstance(mod, nn.BatchNorm2d):  # running statistics as after training, not the initial (0, 1)
                mod.running_mean.copy_(torch.randn_like(mod.running_mean) * 0.1)
                mod.running_var.copy_(torch.rand_like(mod.running_var) + 0.5)
                mod.weight.copy_(torch.rand_like(mod.weight) + 0.5)
                mod.bias.copy_(torch.randn_like(mod.bias) * 0.1)
    xi = torch.rand(2, 3, 32, 32)
    with torch.no_grad():
        yo = r(xi)
    (OUT / "resnet_tiny_opset13.onnx").write_bytes(export(r, (xi,), ["x"], 13))
    np.savez(OUT / "resnet_tiny_io.npz", x=xi.numpy(), y=yo.numpy())
    for f in sorted(OUT.iterdir()):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
