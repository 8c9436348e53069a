"""Global training config with the reference's field names and defaults (reference config.py:1-31).
A tiny attribute-dict replaces the third-party ``easydict`` the reference imports."""


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


config = _AttrDict()
config.dataset = "ms1m_split"
config.embedding_size = 512
config.sample_rate = 1
config.fp16 = True            # reference: fp16 autocast; this backend: fp16 storage (libfedfr_hip.so; bf16 in libfedfr_hip_bf16.so) / fp32 accumulate
config.momentum = 0.9
config.weight_decay = 5e-4
config.lr = 0.05
config.step = [6, 14]
config.num_epoch = 16
config.val_targets = ["agedb_30"]


def lr_step_func(epoch):
    """reference config.py:22-25."""
    return ((epoch + 1) / (4 + 1)) ** 2 if epoch < -1 else 0.1 ** len([m for m in config.step if m - 1 <= epoch])


config.lr_func = lr_step_func
config.com_batch_size = 256
config.public_batch_size = 512
config.HN_threshold = 0.4
config.train_decay = 8
config.mu = 5
config.converter_layer = 1
