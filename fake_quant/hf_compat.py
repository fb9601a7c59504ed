"""Module layouts of newer ``transformers`` releases, presented in the layout the reference walks.

The reference pins transformers 4.46.3 / 4.47.1 (``qwen2vl_environment.yml``); its passes and its ``exam/quant_qwen2vl.py``
dereference that release's attribute layout of ``Qwen2VLForConditionalGeneration``::

    hf.visual                      the vision tower            (qwen2vl_rotation.py:16-60, exam/quant_qwen2vl.py:86-143)
    hf.model                       the TEXT model: .embed_tokens / .layers / .norm   (qwen2vl_rotation.py:232-332)
    hf.lm_head
    hf.config.hidden_size / .intermediate_size / .num_attention_heads / .need_pad    (flat)

From transformers 4.52 on (checked here against the installed 5.15) the same class nests differently: ``hf.model`` is a
``Qwen2VLModel`` holding ``.visual`` AND ``.language_model``, and the text fields of the config moved to
``hf.config.text_config``.  ``legacy_qwen2vl(hf)`` returns a shell with the OLD attribute layout whose children ARE the new model's
sub-modules: every in-place pass of this package (``fuse_qwen2vl_layer_norms``, ``rotate_qwen2vl_model``,
``qwen2vl_add_act_qaunt``, the RTN / GPTQ drivers, the calibration toggles) then edits the real model, and ``shell(...)`` /
``shell.generate(...)`` run it.  A model that already has the old layout is returned unchanged.

The shell is a view for surgery and execution, not a container: it registers nothing (``state_dict`` / ``parameters`` of the real
model are unaffected), so checkpoints are taken from the real model.
"""
from __future__ import annotations

import torch


class _FlatConfig:
    """``config`` of the old layout: text fields readable and writable at the top level (the rotation pass sets
    ``intermediate_size`` and ``need_pad``, qwen2vl_rotation.py:282,308-309); everything else falls through to the real config."""

    def __init__(self, cfg):
        object.__setattr__(self, "_cfg", cfg)
        object.__setattr__(self, "_text", getattr(cfg, "text_config", cfg))

    def __getattr__(self, name):
        cfg, text = object.__getattribute__(self, "_cfg"), object.__getattribute__(self, "_text")
        if hasattr(text, name):
            return getattr(text, name)
        return getattr(cfg, name)

    def __setattr__(self, name, value):
        cfg, text = object.__getattribute__(self, "_cfg"), object.__getattribute__(self, "_text")
        setattr(text if hasattr(text, name) or not hasattr(cfg, name) else cfg, name, value)


class LegacyQwen2VL:
    """Old-layout view of a new-layout ``Qwen2VLForConditionalGeneration`` (see the module docstring)."""

    def __init__(self, hf):
        self.__dict__["hf"] = hf
        self.__dict__["config"] = _FlatConfig(hf.config)

    # the three children the reference dereferences: properties, so that replacing one on the real model stays visible
    @property
    def visual(self):
        return self.hf.model.visual

    @property
    def model(self):
        return self.hf.model.language_model

    @property
    def lm_head(self):
        return self.hf.lm_head

    @lm_head.setter
    def lm_head(self, value):          # exam/quant_qwen2vl.py:37-48 re-creates the head of the 2B model
        self.hf.lm_head = value

    def __getattr__(self, name):       # generate, device, dtype, eval, to ...: the real model's
        return getattr(self.__dict__["hf"], name)

    def __call__(self, *args, **kwargs):
        return self.hf(*args, **kwargs)

    def modules(self):
        return self.hf.modules()

    def named_modules(self, *args, **kwargs):
        return self.hf.named_modules(*args, **kwargs)


def is_new_qwen2vl_layout(hf) -> bool:
    inner = getattr(hf, "model", None)
    return isinstance(hf, torch.nn.Module) and inner is not None and hasattr(inner, "language_model") and hasattr(inner, "visual")


def legacy_qwen2vl(hf):
    """``hf`` itself when it has the layout the reference walks, else the old-layout shell around it."""
    if isinstance(hf, LegacyQwen2VL) or not is_new_qwen2vl_layout(hf):
        return hf
    return LegacyQwen2VL(hf)
