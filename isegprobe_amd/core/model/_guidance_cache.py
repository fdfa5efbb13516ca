"""Cross-click cache of click-independent upsampler work.

In the click loop the reference recomputes, for every click, everything the upsamplers derive from
the guidance image alone (iseg_probe_model.py:113-133): FeatUp-JBU's range/spatial kernels, LoftUp's
Fourier features + first convs + first query projection (42 % of its FLOPs, loftup.py:102-106), LiFT's
image pyramid (LiFT.py:109-111).  The guidance only changes when the predictor's transforms change the
image (new image, new zoom-in ROI), which the predictor already tracks (``is_image_changed``,
base_predictor.py:89).  The predictor opens a ``guidance_scope(token)`` around the network call; inside
it each upsampler memoises its guidance-only intermediates under that token.  Outside a scope (plain
``model(image, points)`` calls, training) nothing is cached, so outputs are identical either way."""
import os
import threading

_state = threading.local()
_DISABLED = bool(os.environ.get("ISEGPROBE_NO_GUIDANCE_CACHE"))  # A/B switch for measurements


class guidance_scope:
    """``with guidance_scope(token):`` -- `token` is any hashable that changes whenever the guidance
    (the normalised image the upsamplers see) changes."""

    def __init__(self, token):
        self.token = token

    def __enter__(self):
        self.prev = getattr(_state, "token", None)
        _state.token = self.token
        return self

    def __exit__(self, *exc):
        _state.token = self.prev


def current_token():
    return None if _DISABLED else getattr(_state, "token", None)


def _tensors(v):
    return list(v) if isinstance(v, (tuple, list)) else [v]


_epoch = [0]


def storage_epoch():
    """Bumped whenever ANY guidance cache allocates, replaces or drops storage.  A captured HIP graph holds raw
    pointers into that storage, so it is only replayable while the epoch it was captured under still stands
    (predictors/base_predictor.py re-captures otherwise)."""
    return _epoch[0]


class _Slot:
    __slots__ = ("token", "data", "fresh")

    def __init__(self):
        self.token, self.data, self.fresh = None, {}, set()


class GuidanceCache:
    """Per-module memo: one slot per geometry (guidance shape, device, packed-weight identity), each valid while its
    token stands.

    Pointer stability: a slot's tensors stay alive while the slot does, and when the token changes but the rebuilt
    tensors have the shapes of the ones they replace, the new values are copied INTO the old storage.  A captured
    HIP graph of the click-dependent part (predictors/base_predictor.py) therefore keeps reading valid addresses
    across zoom-in ROI changes and across geometries that alternate (ZoomIn(skip_clicks=1): click 1 at the image
    size, later clicks at the zoom size; images of different sizes); only the guidance-only producers are re-run
    (eagerly) when the image changes.  Anything that does move storage bumps ``storage_epoch()``."""

    MAX_GEOMETRIES = 8

    def __init__(self):
        from collections import OrderedDict
        self.slots = OrderedDict()

    def get(self, guidance, weights_id, name, build):
        tok = current_token()
        if tok is None:
            return build()
        geom = (tuple(guidance.shape), str(guidance.device), weights_id)
        slot = self.slots.get(geom)
        if slot is None:
            while len(self.slots) >= self.MAX_GEOMETRIES:
                self.slots.popitem(last=False)
                _epoch[0] += 1
            slot = self.slots[geom] = _Slot()
        else:
            self.slots.move_to_end(geom)
        if slot.token != tok:
            slot.token, slot.fresh = tok, set()
        if name not in slot.fresh:
            new = build()
            old = slot.data.get(name)
            same = (old is not None and type(old) is type(new) and len(_tensors(old)) == len(_tensors(new))
                    and all(o.shape == n.shape and o.dtype == n.dtype for o, n in zip(_tensors(old), _tensors(new))))
            if same:  # (a tensor where a tuple was, or tuples of different lengths, replace the entry instead)
                for o, n in zip(_tensors(old), _tensors(new)):
                    o.copy_(n)
            else:
                slot.data[name] = new
                _epoch[0] += 1
            slot.fresh.add(name)
        return slot.data[name]

    def clear(self):
        if self.slots:
            _epoch[0] += 1
        self.slots.clear()
