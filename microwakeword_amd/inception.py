"""Model-module protocol of the reference for Inception (called by model_train_eval):
``model_parameters(parser_nn)``, ``spectrogram_slices_dropped(flags)``, ``model(flags, shape, batch_size)``
— same flags, defaults and help as microwakeword/inception.py:145-209,212-230,232-340, returning the
MI355X-engine-backed :class:`microwakeword_amd.model.Model` (a conv/BN graph context created with
``mww_create_convnet``) instead of a ``tf.keras.Model``.

Dropout (inception.py:330) is active in the train step only; its keep mask comes from the engine's
counter-based generator (``engine.set_option("dropout_seed", s)``) or, for parity tests, from
``engine.set_dropout_mask``."""
from . import layout as _layout
from .model import Model

parse = _layout.parse


def model_parameters(parser_nn):
    """Inception model parameters (names, types, defaults of inception.py:145-209)."""
    parser_nn.add_argument("--cnn1_filters", type=str, default="24", help="Number of filters in the first conv blocks")
    parser_nn.add_argument("--cnn1_kernel_sizes", type=str, default="5", help="Kernel size in time dim of conv blocks")
    parser_nn.add_argument("--cnn1_subspectral_groups", type=str, default="4",
                           help="The number of subspectral groups for normalization")
    parser_nn.add_argument("--cnn2_filters1", type=str, default="10,10,16",
                           help="Number of filters inside of inception block will be multipled by 4 because of "
                                "concatenation of 4 branches")
    parser_nn.add_argument("--cnn2_filters2", type=str, default="10,10,16",
                           help="Number of filters inside of inception block it is used to reduce the dim of cnn2_filters1*4")
    parser_nn.add_argument("--cnn2_kernel_sizes", type=str, default="5,5,5",
                           help="Kernel sizes of conv layers in the inception block")
    parser_nn.add_argument("--cnn2_subspectral_groups", type=str, default="1,1,1",
                           help="The number of subspectral groups for normalization")
    parser_nn.add_argument("--cnn2_dilation", type=str, default="1,1,1", help="Dilation rate")
    parser_nn.add_argument("--dropout", type=float, default=0.2, help="Percentage of data dropped")


def spectrogram_slices_dropped(flags):
    return _layout.inception_slices_dropped(flags)


def model(flags, shape, batch_size, **engine_kwargs):
    return Model(flags, shape, batch_size, layout=_layout.InceptionLayout(flags, int(shape[0])), name="inception", **engine_kwargs)
