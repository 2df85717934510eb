"""Cache keys of the index plans (no GPU needed: the key is made of address, row count and version)."""
import torch


def test_tensor_key_of_inference_mode_tensor():
    """Index tensors made under torch.inference_mode() keep no version counter; the key must still form."""
    from gunet_amd import plan
    with torch.inference_mode():
        t = torch.arange(12, dtype=torch.int32).reshape(6, 2)
        key = plan._tensor_key(t)
    assert key == (t.data_ptr(), 6, -1)
    assert plan._tensor_key(t) == key                       # outside the mode as well
    assert plan._edge_key([t, t]) == (key, key)


def test_tensor_key_changes_with_in_place_edit():
    from gunet_amd import plan
    t = torch.zeros((5, 2), dtype=torch.int32)
    before = plan._tensor_key(t)
    t[0, 0] = 1
    assert plan._tensor_key(t) != before and plan._tensor_key(t)[:2] == before[:2]
