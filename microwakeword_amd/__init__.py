"""MI355X-native (gfx950) train-step engine behind microWakeWord's training interfaces.

Product path = ``libmww_hip.so`` (hand-written HIP kernels + C ABI, ``include/mww.h``); this
package is the host-side mirror of the reference's Python surface for that path.  There is no
CPU fallback: importing works anywhere, running needs the built library and a GPU."""
__all__ = ["native", "layout", "ragged", "data", "model", "mixednet", "train", "parallel"]
