"""allennlp/nn/util.py (subset)."""
import torch


def move_to_device(obj, device):
    from allennlp.common.util import int_to_device

    device = int_to_device(device)
    if isinstance(obj, torch.Tensor):
        return obj if device.type == "cpu" else obj.to(device)
    if isinstance(obj, dict):
        return {k: move_to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, list):
        return [move_to_device(v, device) for v in obj]
    if isinstance(obj, tuple):
        return tuple(move_to_device(v, device) for v in obj)
    return obj


def get_text_field_mask(text_field_tensors, num_wrapping_dims: int = 0, padding_id: int = 0):
    for indexer_tensors in text_field_tensors.values():
        if "mask" in indexer_tensors:
            return indexer_tensors["mask"].bool()
    raise ValueError("no mask in the text field tensors")


def get_lengths_from_binary_sequence_mask(mask):
    return mask.sum(-1)


def get_mask_from_sequence_lengths(sequence_lengths, max_length: int):
    ones = sequence_lengths.new_ones(sequence_lengths.size(0), max_length)
    return sequence_lengths.unsqueeze(1) >= ones.cumsum(dim=1)


def sort_batch_by_length(tensor, sequence_lengths):
    sorted_lengths, perm = sequence_lengths.sort(0, descending=True)
    sorted_tensor = tensor.index_select(0, perm)
    _, restoration = perm.sort(0, descending=False)
    return sorted_tensor, sorted_lengths, restoration, perm


def get_final_encoder_states(encoder_outputs, mask, bidirectional: bool = False):
    last = mask.sum(1).long() - 1
    b, _, d = encoder_outputs.size()
    return encoder_outputs.gather(1, last.view(-1, 1, 1).expand(b, 1, d)).squeeze(1)


def batched_index_select(target, indices, flattened_indices=None):
    b = target.size(0)
    offs = torch.arange(b, device=target.device).view(b, *([1] * (indices.dim() - 1))) * target.size(1)
    flat = target.reshape(-1, target.size(-1))
    return flat.index_select(0, (indices + offs).view(-1)).view(*indices.size(), target.size(-1))


def weighted_sum(matrix, attention):
    if attention.dim() == 2 and matrix.dim() == 3:
        return attention.unsqueeze(1).bmm(matrix).squeeze(1)
    if attention.dim() == 3 and matrix.dim() == 3:
        return attention.bmm(matrix)
    raise ValueError("allennlp stub: weighted_sum supports 2-D / 3-D attention only")
