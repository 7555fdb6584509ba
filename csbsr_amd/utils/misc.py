"""Checkpoint interchange helper.  Same contract as the reference's
``model/utils/misc.py:35-44`` (called from train.py:102, test.py:45,52, build_model.py:98,108): keys saved from an
``nn.DataParallel`` wrapper lose their ``module.`` prefix, then ``len(addition_word)`` leading characters are dropped from every key
(the reference slices by length without checking the text, so a key that does not start with the word is truncated all the same --
kept, because released checkpoints are loaded through exactly that behaviour)."""
from collections import OrderedDict

_DP_PREFIX = "module."


def fix_model_state_dict(state_dict, addition_word=""):
    cut = len(addition_word)

    def rename(key):
        key = key[len(_DP_PREFIX):] if key.startswith(_DP_PREFIX) else key
        return key[cut:]

    return OrderedDict((rename(key), value) for key, value in state_dict.items())
