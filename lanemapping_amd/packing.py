"""Cache of kernel-ready (packed / BN-folded) weights per module, rebuilt when parameters change."""
import torch
import torch.nn as nn


class PackedModule(nn.Module):
    """Subclasses implement ``_pack() -> dict``; ``packed()`` memoises it on the tensors' identities and versions, so
    ``load_state_dict`` / ``.to(device)`` / in-place edits invalidate the cache automatically.

    The key is taken from a cached list of (owner module, slot dict, name) triples instead of walking ``named_modules()``
    on every forward (that walk cost 0.6 ms per call, a third of the host time of a launch sequence); the module tree of an
    inference net does not change after construction, and a replaced tensor object in a slot still changes the key."""

    def _slots(self):
        slots = self.__dict__.get('_packed_slots')
        if slots is None:
            slots = []
            for m in self.modules():
                slots += [(m._parameters, n) for n in m._parameters] + [(m._buffers, n) for n in m._buffers]
            self.__dict__['_packed_slots'] = slots
        return slots

    def param_key(self):
        """Identity + version of every parameter / buffer below this module: what the packed cache - and a captured HIP graph, which
        bakes the packed pointers in - is valid for."""
        return tuple((t.data_ptr(), t._version) for d, n in self._slots() for t in (d[n],) if t is not None)

    def packed(self):
        key = self.param_key()
        cache = self.__dict__.get('_packed_cache')
        if cache is None or cache[0] != key:
            with torch.no_grad():
                P = self._pack()
            # the packing kernels run on the CALLER's stream; another stream that picks the cached dict up must not read the
            # packed weights before they are written: it waits for this event (dropped once it has completed)
            ev = None
            if torch.cuda.is_available() and any(isinstance(v, torch.Tensor) and v.is_cuda for v in P.values()):
                ev = torch.cuda.Event()
                ev.record()
            cache = [key, P, ev, torch.cuda.current_stream().cuda_stream if ev is not None else None]
            self.__dict__['_packed_cache'] = cache
        elif cache[2] is not None and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            # (not while a HIP graph is being captured: event queries invalidate the capture, and TilePipeline captures only after a
            # warm-up run and a device synchronisation, i.e. with every packing kernel complete)
            if cache[2].query():
                cache[2] = None
            elif torch.cuda.current_stream().cuda_stream != cache[3]:
                torch.cuda.current_stream().wait_event(cache[2])
        return cache[1]
