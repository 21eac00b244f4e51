"""Model-module protocol of the reference for MixedNet (called by model_train_eval):
``model_parameters(parser_nn)``, ``spectrogram_slices_dropped(flags)``, ``model(flags, shape, batch_size)``
— same flags and defaults as microwakeword/mixednet.py:43-105,108-129,278-386, returning the
MI355X-engine-backed :class:`microwakeword_amd.model.Model` instead of a ``tf.keras.Model``."""
import logging

from . import layout as _layout
from . import native as _native
from .model import Model

parse = _layout.parse


def model_parameters(parser_nn):
    """MixedNet model parameters (names, types, defaults and help of mixednet.py:43-105)."""
    parser_nn.add_argument("--pointwise_filters", type=str, default="48, 48, 48, 48",
                           help="Number of filters in every MixConv block's pointwise convolution")
    parser_nn.add_argument("--residual_connection", type=str, default="0,0,0,0,0",
                           help="Use a residual connection in each MixConv block")
    parser_nn.add_argument("--repeat_in_block", type=str, default="1,1,1,1",
                           help="Number of repeating conv blocks inside of residual block")
    parser_nn.add_argument("--mixconv_kernel_sizes", type=str, default="[5], [9], [13], [21]",
                           help="Kernel size lists for DepthwiseConv1D in time dim for every MixConv block")
    parser_nn.add_argument("--max_pool", type=int, default=0,
                           help="apply max pool instead of average pool before final convolution and sigmoid activation")
    parser_nn.add_argument("--first_conv_filters", type=int, default=32,
                           help="Number of filters on initial convolution layer. Set to 0 to disable.")
    parser_nn.add_argument("--first_conv_kernel_size", type=int, default="3",
                           help="Temporal kernel size for the initial convolution layer.")
    parser_nn.add_argument("--spatial_attention", type=int, default=0,
                           help="Add a spatial attention layer before the final pooling layer")
    parser_nn.add_argument("--pooled", type=int, default=0,
                           help="Pool the temporal dimension before the final fully connected layer.")
    parser_nn.add_argument("--stride", type=int, default=1,
                           help="Striding in the time dimension of the initial convolution layer")


def spectrogram_slices_dropped(flags):
    return _layout.spectrogram_slices_dropped(flags)


def model(flags, shape, batch_size, **engine_kwargs):
    """Raises ValueError("all input lists have to be the same length") exactly where the reference
    does (mixednet.py:298-305) — note the reference's own default ``--residual_connection`` has five
    entries against four blocks, so default flags need ``--residual_connection "0,0,0,0"``."""
    try:
        m = Model(flags, shape, batch_size, **engine_kwargs)   # specialised MFMA block kernels
        m.kernel_family = "specialised block kernels (MFMA pointwise, csrc/block_launch.hip.h)"
        return m
    except (NotImplementedError, _native.NativeError) as e:
        if isinstance(e, _native.NativeError) and "error -3" not in str(e):   # anything but MWW_ERR_UNSUPPORTED
            raise
        # shapes / options the block kernels do not cover run as a generic conv/BN graph (slower VALU kernels);
        # residual / attention / pooled heads still raise NotImplementedError from the layout below
        lay = _layout.GraphMixedNetLayout(flags, int(shape[0]))
        logging.getLogger("microwakeword_amd").warning("mixednet: %s -> generic graph kernels", e)
        m = Model(flags, shape, batch_size, layout=lay, name="mixednet (generic graph kernels)", **engine_kwargs)
        m.kernel_family = "conv / depthwise graph kernels (%s)" % e
        return m


def kernel_family(flags, frames, lib=None, bf16=False):
    """Which kernels a flag set gets, without building the model or touching a GPU: ("block", "") when every block has a
    specialised MFMA kernel, ("graph", reason) when the model runs on the conv / depthwise graph kernels (about 2.5x the
    step time of a covered shape: the reason names the first thing the shape table does not hold)."""
    try:
        lay = _layout.MixedNetLayout(flags, int(frames))
    except NotImplementedError as e:
        return "graph", str(e)
    ok, why = (lib or _native.NativeLib.get()).block_kernels_cover(bf16=bf16, **lay.engine_args(1))
    return ("block", "") if ok else ("graph", why)
