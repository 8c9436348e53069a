"""Reference-compatible ``backbones`` namespace (reference backbones/__init__.py:1): the FL code looks
models up with ``eval("backbones.{}".format(args.network))`` (client.py:133, server.py:83)."""
from .iresnet import iresnet18, iresnet34, iresnet50, iresnet100, iresnet200, IResNet, IBasicBlock  # noqa: F401

from .sphnet import sphere, sphnet  # noqa: E402,F401  (reference backbones/__init__.py exposes sphnet the same way)
