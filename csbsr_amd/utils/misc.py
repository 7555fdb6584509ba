"""Checkpoint interchange helpers: model/utils/misc.py:35-44."""
from collections import OrderedDict


def fix_model_state_dict(state_dict, addition_word=''):
    """strip nn.DataParallel's ``module.`` prefix (and an optional leading ``addition_word``) from every key"""
    new_state_dict = OrderedDict()
    for k, v in state_dict.items():
        name = k
        if name.startswith('module.'):
            name = name[7:]
        if len(addition_word) != 0:
            name = name[len(addition_word):]
        new_state_dict[name] = v
    return new_state_dict
