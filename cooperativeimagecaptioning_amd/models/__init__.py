"""models.setup / models.load / models.AlternatingJointModel — the reference's factory API
(models/__init__.py:14-52)."""
import os

import torch

from ..misc import utils
from .AttModel import Att2in2Model, AttModel, Att2in2Core, Attention  # noqa: F401
from .VSEFCModel import VSEFCModel  # noqa: F401
from .FCModel import FCModel  # noqa: F401

__all__ = ['setup', 'load', 'AlternatingJointModel']


def setup(opt, model_name, model_type='caption_model'):
    if model_type == 'caption_model':
        if model_name == 'att2in2':
            return Att2in2Model(opt)
        if model_name == 'fc':
            return FCModel(opt)
        raise Exception("Caption model not supported: {}".format(model_name))
    elif model_type == 'vse_model':
        if model_name == 'fc':
            return VSEFCModel(opt)
        raise Exception("VSE model not supported: {}".format(model_name))
    raise Exception("model_type not supported: {}".format(model_type))


def load(model, opt, iteration=None):
    """models/__init__.py:35-50: continue from <start_from>/model[-<iteration>].pth (tensors only)."""
    if vars(opt).get('start_from', None) is not None:
        assert os.path.isdir(opt.start_from), " %s must be a a path" % opt.start_from
        # this implementation's checkpoints carry infos_<id>.json (train.checkpoint); a reference checkpoint directory
        # carries infos_<id>.pkl, which is accepted as a marker but never unpickled
        assert any(os.path.isfile(os.path.join(opt.start_from, "infos_" + opt.id + ext)) for ext in ('.json', '.pkl')), \
            "infos file does not exist in path %s" % opt.start_from
        name = 'model-' + iteration + '.pth' if iteration else 'model.pth'
        sd = torch.load(os.path.join(opt.start_from, name), map_location='cpu', weights_only=True)
        utils.load_state_dict(model, sd)


from .AlternatingJointModel import AlternatingJointModel  # noqa: E402,F401
