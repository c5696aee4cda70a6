"""`onnxsim.simplify` stand-in (no onnx-simplifier in the image): returns the model unchanged with check = True. The front-end treats
simplification as best effort — a failed simplification keeps the original model (onnx.py:49-57) — so an unsimplified model is an input
it accepts by construction; the committed fixtures are torch.onnx.export output (constants already folded by the exporter)."""


def simplify(model, *args, **kwargs):
    return model, True
