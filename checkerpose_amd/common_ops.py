"""The two live helpers of reference checkerpose/common_ops.py (config plumbing used by train.py:84,130,221);
the sigmoid/threshold helpers there are dead code."""


def get_batch_size(second_dataset_ratio, batch_size):
    """common_ops.py:43-46"""
    batch_size_2_dataset = int(batch_size * second_dataset_ratio)
    batch_size_1_dataset = batch_size - batch_size_2_dataset
    return batch_size_1_dataset, batch_size_2_dataset


def from_dim_str_to_tuple(src_str):
    """common_ops.py:50-56: '1024_256_32' -> (1024, 256, 32); None -> None"""
    if src_str is None:
        return None
    return tuple(int(dim) for dim in src_str.split("_"))
