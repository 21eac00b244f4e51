"""Shared by tools/keras_weights_to_npz.py and tools/npz_to_keras_weights.py (run where TensorFlow / Keras 3 is installed; nothing
here imports microwakeword_amd).

``microwakeword_amd.model.Model.get_weights / set_weights / save_weights`` keep the variables in the order the reference's
builders CREATE them (mixednet.py:278-386, inception.py:232-340; checked by executing those files: oracle/ref_model_shim.py).
A Keras functional model lists ``model.weights`` by layer, and its layers by graph depth (ties by traversal order): for a
sequential MixedNet - MixConv groups included - that is the creation order; with a residual branch (created before the block it
is added to, shallower in the graph) or Inception's three branches it is NOT.  So the tools do not trust positions: every layer
constructed while the reference's ``model()`` runs is logged, and the variables each layer owns itself (kernel, bias | gamma,
beta, moving_mean, moving_variance) are listed in that order.  Layers are called right after they are constructed in both
builders (a Stream's cell and a SubSpectralNormalization's BatchNormalization are constructed just before / inside their
wrapper and built when it is called), so the log order of the variable-owning layers IS the variable creation order."""
import contextlib


@contextlib.contextmanager
def layer_creation_log(layer_base):
    """Every ``layer_base`` subclass instance constructed inside the block is appended to the yielded list, once."""
    created, seen = [], set()
    original = layer_base.__init__

    def recording_init(self, *args, **kwargs):
        original(self, *args, **kwargs)
        if id(self) not in seen:
            seen.add(id(self))
            created.append(self)

    layer_base.__init__ = recording_init
    try:
        yield created
    finally:
        layer_base.__init__ = original


def own_variables(layer):
    """the variables a Keras 3 layer tracks itself (not its sublayers'): trainable first, in creation order"""
    return list(getattr(layer, "_trainable_variables", [])) + list(getattr(layer, "_non_trainable_variables", []))


def creation_permutation(model_weights, created_layers):
    """-> perm with ``[model_weights[i] for i in perm]`` in creation order; raises if the log does not cover every weight exactly once"""
    index = {id(v): i for i, v in enumerate(model_weights)}
    perm, used = [], set()
    for layer in created_layers:
        for v in own_variables(layer):
            i = index.get(id(v))
            if i is not None and i not in used:        # (seed-generator state and the like are not in model.weights)
                used.add(i)
                perm.append(i)
    if len(perm) != len(model_weights):
        missing = [getattr(model_weights[i], "path", getattr(model_weights[i], "name", i)) for i in range(len(model_weights)) if i not in used]
        raise RuntimeError("layer creation log covers %d of %d model weights; not owned by a logged layer: %s" % (len(perm), len(model_weights), missing[:6]))
    return perm
