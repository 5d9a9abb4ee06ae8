"""Cache of kernel-ready (packed / transposed / concatenated) weights derived from nn.Parameters.

Entries are rebuilt when any source tensor changes storage or is modified in place
(load_state_dict, optimizer steps, .to(device)), detected through (data_ptr, _version)."""


class WeightCache:
    def __init__(self):
        self._store = {}

    def get(self, key, sources, build):
        sig = tuple((t.data_ptr(), t._version, t.device) for t in sources)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = build()
        self._store[key] = (sig, val)
        return val

    def clear(self):
        self._store.clear()
