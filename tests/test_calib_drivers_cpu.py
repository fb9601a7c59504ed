"""Calibration drivers (reference quant_utils.py:877-1031): the open -> ... -> last -> close ->
quant protocol as seen from inside ``generate``, with stand-ins for the VLMEvalKit / HF objects."""
import json
import types

import pandas as pd
import pytest
import torch

from fake_quant import quant_utils as qu

torch.set_grad_enabled(False)


class Args:
    skip_names = []
    dataset_name = "toy"
    calib_num = 3
    calib_mode = "v1"


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1, self.fc2 = torch.nn.Linear(16, 16), torch.nn.Linear(16, 8)

    def forward(self, x):
        return self.fc2(self.fc1(x))


def _wrapped_net():
    net = Net()
    qu.add_actquant(net)
    for w in qu.find_qlayers(net, [qu.ActQuantWrapper]).values():
        w.quantizer.configure(bits=8, sym=True, static=True)
    return net


def _flags(net):
    w = next(iter(qu.find_qlayers(net, [qu.ActQuantWrapper]).values()))
    return (w.quantizer.calibrate, w.quantizer.last_calibrate, w.quantizer.quant)


class VlmWrapper:
    """VLMEvalKit-style object: .model, .generate(message=, dataset=), kwargs dict."""

    def __init__(self, kwargs_attr):
        self.model = _wrapped_net()
        setattr(self, kwargs_attr, {"max_new_tokens": 128})
        self.kwargs_attr = kwargs_attr
        self.calls = []

    def generate(self, message, dataset):
        self.model(torch.full((2, 16), float(message)))
        self.calls.append((message, dataset, getattr(self, self.kwargs_attr)["max_new_tokens"], _flags(self.model)))


class Dataset:
    def __init__(self, n):
        self.data = pd.DataFrame({"v": list(range(1, n + 1))})

    def build_prompt(self, record):
        return int(record["v"])


@pytest.mark.parametrize("driver,attr", [(qu.calib_vqa_plus, "kwargs"), (qu.calib_qwen2vl_plus, "generate_kwargs")])
def test_vlmeval_drivers_follow_the_protocol(driver, attr):
    m = VlmWrapper(attr)
    driver(m, Args(), Dataset(10), calib_num=3)                # step = ceil(10/3) = 4 -> records 0, 4, 8
    assert [c[0] for c in m.calls] == [1, 5, 9]
    assert [c[2] for c in m.calls] == [20, 20, 1]
    assert [c[3] for c in m.calls] == [(True, False, False), (True, False, False), (True, True, False)]
    assert _flags(m.model) == (False, True, True)      # close leaves last_calibrate set, as upstream
    if attr == "kwargs":
        assert m.kwargs == {}
    else:
        assert m.generate_kwargs["max_new_tokens"] == 128
    scale = next(iter(qu.find_qlayers(m.model, [qu.ActQuantWrapper]).values())).quantizer.quantizer.scale
    assert float(scale) == pytest.approx(9.0 / 127.0)          # running max over the three batches


class Tokenizer:
    eod_id = 7
    padding_side = "right"
    pad_token_id = None

    def __call__(self, questions, return_tensors, padding):
        n = max(len(q) for q in questions)
        ids = torch.tensor([[self.pad_token_id] * (n - len(q)) + [ord(c) % 50 for c in q] for q in questions])
        return types.SimpleNamespace(input_ids=ids, attention_mask=(ids != self.pad_token_id).long())


class HfModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = _wrapped_net()
        self.calls = []

    def generate(self, input_ids, attention_mask, max_new_tokens, **kw):
        assert kw["pad_token_id"] == kw["eos_token_id"] == 7 and kw["do_sample"] is False
        self.net(torch.ones(input_ids.shape[0], 16))
        self.calls.append((tuple(input_ids.shape), max_new_tokens, _flags(self.net)))


@pytest.mark.parametrize("mode,records,expect_tokens", [("v1", 12, [10, 10, 1]), ("v2", 14, [10, 10, 10, 1])])
def test_calib_vqa_jsonl_driver(tmp_path, monkeypatch, mode, records, expect_tokens):
    train = tmp_path / "train.jsonl"
    with open(train, "w") as fh:
        for i in range(records):
            fh.write(json.dumps({"image": f"img{i}.jpg", "question": "what?" + "?" * i, "question_id": i,
                                 "answer": "yes"}) + "\n")
    monkeypatch.setitem(qu.ds_collections, "toy", {"train": str(train), "test": str(train), "metric": None,
                                                   "max_new_tokens": 10})
    args = Args()
    args.calib_mode = mode
    model = HfModel()
    qu.calib_vqa(model, Tokenizer(), args, "toy", batch_size=2, num_workers=0)
    # v1: batches 0,1,2.  v2: 7 batches, step 7 // 3 = 2 -> batches 0,2,4,6; "last" is the first
    # sampled batch with idx + step > n_batches (upstream's rule, :936-939)
    assert [c[1] for c in model.calls] == expect_tokens
    assert [c[2] for c in model.calls] == [(True, False, False)] * (len(expect_tokens) - 1) + [(True, True, False)]
    assert all(c[0][0] == 2 for c in model.calls)
    assert _flags(model.net) == (False, True, True)
    with pytest.raises(ValueError):
        args.calib_mode = "v3"
        qu.calib_vqa(HfModel(), Tokenizer(), args, "toy", batch_size=2, num_workers=0)


def test_vqa_dataset_few_shot_prompt(tmp_path):
    path = tmp_path / "d.jsonl"
    with open(path, "w") as fh:
        for i in range(3):
            fh.write(json.dumps({"image": f"{i}.png", "question": f"q{i}", "question_id": 100 + i, "answer": f"a{i}"}) + "\n")
    ds = qu.VQADataset(str(path), str(path), "<img>{}</img>{} Answer:", few_shot=2)
    item = ds[1]
    assert item["question_id"] == 101 and item["annotation"] == "a1"
    assert item["question"].endswith("<img>1.png</img>q1 Answer:") and item["question"].count("Answer:") == 3
    assert len(ds) == 3 and set(qu.ds_collections["docvqa_val"]) == {"train", "test", "annotation", "metric", "max_new_tokens"}
