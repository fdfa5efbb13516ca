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


class GuidanceCache:
    """Per-module memo, valid while (token, guidance shape/device, packed-weight identity) stay the same.

    Pointer stability: when the token changes but the rebuilt tensors have the shapes of the ones they
    replace, the new values are copied INTO the old storage.  A captured HIP graph of the click-dependent
    part (predictors/base_predictor.py) therefore keeps reading valid addresses across zoom-in ROI changes;
    only the guidance-only producers are re-run (eagerly) when the image changes."""

    def __init__(self):
        self.key, self.data, self.fresh = None, {}, set()

    def get(self, guidance, weights_id, name, build):
        tok = current_token()
        if tok is None:
            return build()
        key = (tok, tuple(guidance.shape), str(guidance.device), weights_id)
        if key != self.key:
            if self.key is not None and key[1:] != self.key[1:]:
                self.data = {}  # other geometry / weights: nothing to reuse
            self.key, self.fresh = key, set()
        if name not in self.fresh:
            new = build()
            old = self.data.get(name)
            if old is not None and all(o.shape == n.shape and o.dtype == n.dtype for o, n in zip(_tensors(old), _tensors(new))):
                for o, n in zip(_tensors(old), _tensors(new)):
                    o.copy_(n)
            else:
                self.data[name] = new
            self.fresh.add(name)
        return self.data[name]

    def clear(self):
        self.key, self.data, self.fresh = None, {}, set()
