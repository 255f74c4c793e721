"""Cache of kernel-ready (packed / BN-folded) weights per module, rebuilt when parameters change."""
import torch
import torch.nn as nn


class PackedModule(nn.Module):
    """Subclasses implement ``_pack() -> dict``; ``packed()`` memoises it on the tensors' versions,
    so ``load_state_dict`` / ``.to(device)`` / in-place edits invalidate the cache automatically."""

    def packed(self):
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cache = self.__dict__.get('_packed_cache')
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, self._pack())
            self.__dict__['_packed_cache'] = cache
        return cache[1]
