"""The activation workspace is bounded (VERDICT r4 "What's weak" 9): one grow-only buffer per (device, K_pad, layout),
sliced per call -- not one buffer per distinct row count kept forever."""
import torch

from mquant_amd import engine, ops


def test_workspace_holds_one_buffer_per_k_pad_whatever_the_row_counts():
    ws = engine.Workspace()
    dev = torch.device("cpu")
    sizes = list(range(1, 801, 4))                            # 200 distinct prompt lengths
    for M in sizes:
        for K_pad in (3584, 19968):
            a = ws.act(dev, M, K_pad)
            assert isinstance(a, ops.TiledAct) and a.M == M and a.K_pad == K_pad
            assert a.data.numel() == ops.ceil_to(M, 16) * K_pad and a.data.is_contiguous()
        assert ws.x0(dev, M).shape == (M,)
    top = ops.ceil_to(max(sizes), 16)
    assert ws.nbytes() == top * 3584 + top * 19968 + 4 * max(sizes)
    assert len(ws._a) == 2 and not ws._pinned
    # a smaller request afterwards is a prefix of the same storage: nothing new is allocated
    a = ws.act(dev, 16, 3584)
    b = ws.act(dev, 640, 3584)
    assert a.data.data_ptr() == b.data.data_ptr()


def test_a_prefix_of_the_tiled_image_is_the_tiled_image_of_fewer_rows():
    rows = torch.randint(-128, 128, (77, 256), dtype=torch.int8)
    full = ops.TiledAct.from_rows(rows)
    for M in (1, 16, 17, 48, 77):
        part = ops.TiledAct(full.data.reshape(-1)[: ops.ceil_to(M, 16) * 256].view(-1, 4, 64, 16), M, 256)
        assert torch.equal(part.to_rows(), rows[:M])


def test_outgrown_buffers_are_pinned_only_when_a_capture_saw_them(monkeypatch):
    ws = engine.Workspace()
    dev = torch.device("cpu")
    a = ws.act(dev, 32, 128)
    ws.act(dev, 64, 128)                                       # grows; nothing was captured: the old buffer is dropped
    assert not ws._pinned
    monkeypatch.setattr(engine.Workspace, "_capturing", staticmethod(lambda: True))
    b = ws.act(dev, 64, 128)                                   # handed out "during capture"
    monkeypatch.setattr(engine.Workspace, "_capturing", staticmethod(lambda: False))
    c = ws.act(dev, 256, 128)                                  # grows: the captured buffer must stay alive
    assert len(ws._pinned) == 1 and ws._pinned[0].data_ptr() == b.data.data_ptr() != c.data.data_ptr()
    assert ws.nbytes() == 256 * 128 + 64 * 128
    del a
