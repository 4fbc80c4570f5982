"""Host-side mirror of the reference's Python ``SentencePieceProcessor`` for the
encode -> ids path, over the C ABI (``include/spmx.h``).

Same names, argument meaning and error behaviour as the reference wrapper
(python/src/sentencepiece/__init__.py:471-562 ``Encode``; sentencepiece.i:
439-446 ``_EncodeAsIdsBatch``; :138-145 ``RewriteIds``), restricted to
``out_type=int`` and deterministic encoding.  Everything that computes runs in
libspmx.so on the GPU; this file only marshals buffers.

Besides the list-of-str form the processor takes the packed form the engine
works on -- one ``uint8`` text blob + ``uint64`` offsets (host numpy arrays or
device torch tensors) -- and returns CSR ``(ids, id_offsets)``.
"""
import ctypes as C
import os

import numpy as np

from . import _capi

_PYEXT = [False, None]


def _pyext():
    """csrc/pyext.c (built by csrc/Makefile next to libspmx.so), or None."""
    if _PYEXT[0] is False:
        try:
            from . import _spmx_py
            _PYEXT[1] = _spmx_py
        except ImportError:
            _PYEXT[1] = None
        _PYEXT[0] = True
    return _PYEXT[1]

_OK = 0
_RESOURCE_EXHAUSTED = 8


class SentencePieceProcessor:
    def __init__(self, model_file=None, model_proto=None, out_type=int, add_bos=False, add_eos=False,
                 reverse=False, emit_unk_piece=False, enable_sampling=False, nbest_size=-1, alpha=0.1,
                 num_threads=-1, device=0, _lib=None):
        self._lib = _lib if _lib is not None else _capi.lib()   # _lib: test seam (tests/emu), never set by the product
        self._h = None
        self._device = device
        self._out_type = out_type
        self._add_bos, self._add_eos, self._reverse = add_bos, add_eos, reverse
        self._enable_sampling = enable_sampling
        self._nbest_size, self._alpha = nbest_size, alpha
        self._emit_unk_piece = emit_unk_piece
        self._extra = ""          # SetEncodeExtraOptions string
        self._applied = ""        # option string currently compiled into the handle
        if model_file or model_proto:
            self.Load(model_file=model_file, model_proto=model_proto)

    # ------------------------------------------------------------- load ----
    def Load(self, model_file=None, model_proto=None):
        """Load / LoadFromSerializedProto (src/sentencepiece_processor.h:245, :261)."""
        if model_proto is None:
            if model_file is None:
                raise RuntimeError("model_file or model_proto must be given")
            try:
                with open(model_file, "rb") as f:
                    model_proto = f.read()
            except OSError:
                raise OSError('Not found: "%s": No such file or directory' % model_file)
        self._close()
        h = C.c_void_p()
        rc = self._lib.spmx_create(model_proto, len(model_proto), self._device, C.byref(h))
        if rc != _OK:
            raise RuntimeError(self._lib.spmx_last_error(None).decode("utf-8", "replace"))
        self._h = h
        self._pid = os.getpid()
        self._extra = self._applied = ""
        self._model_proto = bytes(model_proto)
        return True

    LoadFromSerializedProto = lambda self, proto: self.Load(model_proto=proto)  # noqa: E731
    load = Load
    load_from_serialized_proto = LoadFromSerializedProto

    def LoadFromFile(self, arg):
        """``LoadFromFile`` (python/src/sentencepiece/__init__.py:315-316)."""
        return self.Load(model_file=arg)

    load_from_file = LoadFromFile

    def serialized_model_proto(self):
        """The ModelProto the processor was loaded from (src/sentencepiece_processor.h:667)."""
        self._need()
        return self._model_proto

    def _close(self):
        if self._h:
            # A handle belongs to the process that made it: a forked child (a multiprocessing pool started while a
            # processor is alive) inherits the object, not the HIP context behind it -- its collector must not call into the
            # runtime (that aborts the child, and a pool then waits for ever for the worker it lost).
            if getattr(self, "_pid", None) == os.getpid():
                self._lib.spmx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _OK:
            raise RuntimeError(self._lib.spmx_last_error(self._h).decode("utf-8", "replace"))

    def _need(self):
        if not self._h:
            raise RuntimeError("Model is not initialized.")   # sentencepiece_processor.cc:293-299

    # ----------------------------------------------------- configuration ---
    def SetEncodeExtraOptions(self, extra_option):
        """bos:eos:reverse in any order (src/sentencepiece_processor.cc:283-291, :1067-1101)."""
        self._need()
        self._check(self._lib.spmx_set_encode_extra_options(self._h, extra_option.encode()))
        self._extra = self._applied = extra_option
        return True

    def SetDecodeExtraOptions(self, extra_option):
        """bos:eos:reverse in any order, applied to the pieces of every later Decode before they are turned into text
        (src/sentencepiece_processor.h:270, .cc:288-291, :819)."""
        self._need()
        self._check(self._lib.spmx_set_decode_extra_options(self._h, extra_option.encode()))
        return True

    def _apply(self, add_bos, add_eos, reverse):
        # RewriteIds (sentencepiece.i:138-145) runs after the processor's own
        # extra options: reverse, then bos in front, then eos at the back --
        # the same net effect as appending "reverse:bos:eos" to the option list.
        opts = [o for o in (self._extra,) if o]
        if reverse:
            opts.append("reverse")
        if add_bos:
            opts.append("bos")
        if add_eos:
            opts.append("eos")
        want = ":".join(opts)
        if want != self._applied:
            self._check(self._lib.spmx_set_encode_extra_options(self._h, want.encode()))
            self._applied = want

    def SetVocabulary(self, valid_vocab):
        self._need()
        bs = [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in valid_vocab]
        arr = (C.c_char_p * len(bs))(*bs)
        lens = (C.c_uint64 * len(bs))(*[len(b) for b in bs])
        self._check(self._lib.spmx_set_vocabulary(self._h, arr, lens, len(bs)))
        return True

    def ResetVocabulary(self):
        self._need()
        self._check(self._lib.spmx_reset_vocabulary(self._h))
        return True

    # ------------------------------------------------------- vocabulary ----
    def GetPieceSize(self):
        self._need()
        return self._lib.spmx_piece_size(self._h)

    piece_size = vocab_size = __len__ = GetPieceSize

    def PieceToId(self, piece):
        self._need()
        b = piece.encode("utf-8") if isinstance(piece, str) else bytes(piece)
        return self._lib.spmx_piece_to_id(self._h, b, len(b))

    piece_to_id = PieceToId

    def IdToPiece(self, id):
        self._need()
        n = self._lib.spmx_id_to_piece(self._h, id, None, 0)
        if n < 0:
            raise IndexError("piece id is out of range.")
        buf = C.create_string_buffer(int(n) + 1)
        self._lib.spmx_id_to_piece(self._h, id, buf, n)
        return buf.raw[:n].decode("utf-8", "replace")

    id_to_piece = IdToPiece

    def unk_id(self):
        self._need()
        return self._lib.spmx_unk_id(self._h)

    def _type(self, id):
        self._need()
        return self._lib.spmx_piece_type(self._h, id)

    def IsUnknown(self, id):
        return self._type(id) == 2

    def IsControl(self, id):
        return self._type(id) == 3

    def IsUnused(self, id):
        return self._type(id) == 5

    def IsByte(self, id):
        return self._type(id) == 6

    is_unknown, is_control, is_unused, is_byte = IsUnknown, IsControl, IsUnused, IsByte

    def bos_id(self):
        self._need()
        return self._lib.spmx_bos_id(self._h)

    def eos_id(self):
        self._need()
        return self._lib.spmx_eos_id(self._h)

    def pad_id(self):
        self._need()
        return self._lib.spmx_pad_id(self._h)

    def unk_piece(self):
        """trainer_spec.unk_piece (the string the ``unk_piece`` extra option writes)."""
        self._need()
        n = self._lib.spmx_unk_piece(self._h, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        self._lib.spmx_unk_piece(self._h, buf, n)
        return buf.raw[:n].decode("utf-8", "replace")

    def model_type(self):
        self._need()
        return self._lib.spmx_model_type(self._h)

    # ----------------------------------------------------------- encode ----
    def Encode(self, input, out_type=None, add_bos=None, add_eos=None, reverse=None, emit_unk_piece=None,
               enable_sampling=None, nbest_size=None, alpha=None, num_threads=None):
        """str -> list[int]; list[str] -> list[list[int]] (``_EncodeAsIdsBatch``).

        ``num_threads`` is accepted and ignored: the batch is one GPU launch
        sequence, not a host thread pool."""
        self._need()
        out_type = self._out_type if out_type is None else out_type
        if (self._enable_sampling if enable_sampling is None else enable_sampling):
            if out_type is not int:
                # _SampleEncodeAsPieces / ...AsSerializedProto / ...AsImmutableProto (sentencepiece.i): the drawn
                # segmentation with its pieces, surfaces and byte ranges (spmx_sample_encode_batch_spans)
                nb = int(self._nbest_size if nbest_size is None else nbest_size)
                al = float(self._alpha if alpha is None else alpha)
                if out_type is str or out_type == "str":
                    self._apply(self._add_bos if add_bos is None else add_bos, self._add_eos if add_eos is None else add_eos,
                                self._reverse if reverse is None else reverse)
                    try:
                        single = not isinstance(input, list)
                        rows = self.SampleEncodeAsPieces([input] if single else input, nb, al, _keep_options=True)
                    finally:
                        self._apply(False, False, False)
                    emit = self._emit_unk_piece if emit_unk_piece is None else emit_unk_piece
                    if emit:
                        unk = self.IdToPiece(self.unk_id())
                        rows = [[unk if self.PieceToId(p) == self.unk_id() else p for p in row] for row in rows]
                    return rows[0] if single else rows
                if any([add_bos, add_eos, reverse, emit_unk_piece]):     # sentencepiece.i:177-186
                    raise NotImplementedError("add_bos, add_eos, reverse, and emit_unk_piece is not supported in proto API")
                if out_type == "serialized_proto":
                    return self.SampleEncodeAsSerializedProto(input, nb, al)
                if out_type == "immutable_proto":
                    from . import spt_proto
                    single, raw, _, _ = self._pack_items(input)
                    if not hasattr(self, "_sample_calls"):
                        self._sample_calls = int.from_bytes(os.urandom(7), "little")
                    self._sample_calls += 1
                    rows = self.SampleEncodeAsSentencePieceText(raw, nb, al, seed=self._sample_calls)
                    out = [spt_proto.ImmutableSentencePieceText(r, [(p, t, sf if sf is not None else b"", b, e) for p, t, sf, b, e in rows[i]],
                                                               spt_proto.serialize(r, rows[i])) for i, r in enumerate(raw)]
                    return out[0] if single else out
                raise RuntimeError("unknown out_type=%r" % (out_type,))
            # _SampleEncodeAsIds (sentencepiece.i): SampleEncode + RewriteIds; the draws are keyed by a fresh seed per call
            # (a per-process counter started from os.urandom: a fresh process does not replay the previous one's draws --
            # the reference seeds its generator from std::random_device, src/util.cc:202-204)
            if not hasattr(self, "_sample_calls"):
                self._sample_calls = int.from_bytes(os.urandom(7), "little")
            self._sample_calls += 1
            self._apply(self._add_bos if add_bos is None else add_bos, self._add_eos if add_eos is None else add_eos,
                        self._reverse if reverse is None else reverse)
            single = not isinstance(input, list)
            items = [input] if single else input
            bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in items]
            offs = np.zeros(len(bs) + 1, dtype=np.uint64)
            if bs:
                np.cumsum([len(b) for b in bs], out=offs[1:])
            ids, io = self._csr_call(self._lib.spmx_sample_encode_batch, np.frombuffer(b"".join(bs), dtype=np.uint8), offs,
                                     int(self._nbest_size if nbest_size is None else nbest_size),
                                     C.c_float(self._alpha if alpha is None else alpha), int(self._sample_calls))
            io = io.astype(np.int64)
            out = [ids[io[i]:io[i + 1]].tolist() for i in range(len(bs))]
            return out[0] if single else out
        if out_type is str or out_type == "str":
            # _EncodeAsPiecesBatch (sentencepiece.i:448-456): EncodeAsPieces per element, then RewriteIds on the
            # piece lists (:147-164): reverse, bos piece in front, eos piece at the back, unknown pieces -> unk piece
            single = not isinstance(input, list)
            rows = self.EncodeAsPieces([input] if single else input)
            a_bos = self._add_bos if add_bos is None else add_bos
            a_eos = self._add_eos if add_eos is None else add_eos
            rev = self._reverse if reverse is None else reverse
            emit = self._emit_unk_piece if emit_unk_piece is None else emit_unk_piece
            if a_bos or a_eos or rev or emit:
                unk = self.IdToPiece(self.unk_id())
                for row in rows:
                    if rev:
                        row.reverse()
                    if a_bos:
                        row.insert(0, self.IdToPiece(self.bos_id()))
                    if a_eos:
                        row.append(self.IdToPiece(self.eos_id()))
                    if emit:
                        row[:] = [unk if self.PieceToId(p) == self.unk_id() else p for p in row]
            return rows[0] if single else rows
        if out_type == "immutable_proto":
            if any([add_bos, add_eos, reverse, emit_unk_piece]):     # sentencepiece.i:177-186
                raise NotImplementedError("add_bos, add_eos, reverse, and emit_unk_piece is not supported in proto API")
            return self.EncodeAsImmutableProto(input)
        if out_type == "serialized_proto":
            if any([add_bos, add_eos, reverse, emit_unk_piece]):     # sentencepiece.i:166-175
                raise NotImplementedError("add_bos, add_eos, reverse, and emit_unk_piece is not supported in proto API")
            return self.EncodeAsSerializedProto(input)
        if out_type is not int:
            raise NotImplementedError("out_type int, str and 'serialized_proto' are on the device path")
        self._apply(self._add_bos if add_bos is None else add_bos,
                    self._add_eos if add_eos is None else add_eos,
                    self._reverse if reverse is None else reverse)
        single = not isinstance(input, list)
        items = [input] if single else input
        out = self._encode_items(items, as_lists=True)
        return out[0] if single else out

    def _encode_items(self, items, as_lists):
        """list[str | bytes] -> list[list[int]] (``as_lists``) or the CSR ``(ids int32[total], id_offsets uint64[n + 1])``.
        The str -> (pointer, length) views and the CSR -> lists loops run in C (csrc/pyext.c, the GIL released around the
        device call) -- the loops the reference's SWIG layer has in C++ (sentencepiece.i:439-446); without the extension
        module (not built) the same is done in Python."""
        ext = _pyext()
        if ext is not None and all(isinstance(s, (str, bytes)) for s in items[:1]):
            fn = C.cast(self._lib.spmx_encode_batch_views, C.c_void_p).value
            h = self._h.value if isinstance(self._h, C.c_void_p) else int(self._h)
            p_ids, p_off, n, total = ext.encode_views(fn, h, items)
            if p_ids == 0 and p_off == 0 and n == 0 and len(items):
                self._check(int(total))
            try:
                if as_lists:
                    return ext.csr_to_lists(p_ids, p_off, n)
                io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
                ids = (np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(int(total),)).copy()
                       if total else np.zeros(0, dtype=np.int32))
                return ids, io
            finally:
                self._lib.spmx_free(p_ids)
                self._lib.spmx_free(p_off)
        bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in items]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            np.cumsum([len(b) for b in bs], out=offs[1:])
        text = np.frombuffer(b"".join(bs), dtype=np.uint8)
        ids, io = self._encode_host(text, offs)
        if not as_lists:
            return ids, io
        io = io.astype(np.int64)
        return [ids[io[i]:io[i + 1]].tolist() for i in range(len(bs))]

    def EncodeAsArrays(self, input, add_bos=False, add_eos=False, reverse=False):
        """list[str] -> ``(ids int32[total], id_offsets uint64[n + 1])``: the batch as ONE flat array pair instead of n
        Python lists of Python ints (which cost more host time than the whole device path: ~25 ns per int object).
        Sentence i is ``ids[id_offsets[i]:id_offsets[i + 1]]``."""
        self._need()
        self._apply(add_bos, add_eos, reverse)
        return self._encode_items(list(input), as_lists=False)

    encode = Encode

    def EncodeAsIds(self, input, **kw):
        return self.Encode(input, out_type=int, **kw)

    encode_as_ids = EncodeAsIds

    def EncodePacked(self, text, offsets, add_bos=False, add_eos=False, reverse=False):
        """Packed host arrays -> CSR host arrays ``(ids int32, id_offsets uint64)``."""
        self._need()
        self._apply(add_bos, add_eos, reverse)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return self._encode_host(text, offsets)

    def EncodePackedEx(self, text, offsets, add_bos=False, add_eos=False, reverse=False):
        """As EncodePacked, plus the per-sentence status bytes (util::StatusCode numbers, 0 = OK; a failed sentence has
        no ids, the others are unaffected -- what the reference's batch form does with a failing element) and their
        count: ``(ids, id_offsets, status uint8[n], n_failed)``."""
        self._need()
        self._apply(add_bos, add_eos, reverse)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        p_ids, p_off, p_st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        failed = C.c_uint64(0)
        tp = text.ctypes.data if len(text) else None
        self._check(self._lib.spmx_encode_batch_ex(self._h, tp, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off),
                                                   C.byref(p_st), C.byref(failed)))
        try:
            io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(io[n])
            ids = (np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
            st = (np.ctypeslib.as_array(C.cast(p_st, C.POINTER(C.c_uint8)), shape=(n,)).copy()
                  if n else np.zeros(0, dtype=np.uint8))
        finally:
            for p in (p_ids, p_off, p_st):
                self._lib.spmx_free(p)
        return ids, io, st, int(failed.value)

    def _encode_host(self, text, offs):
        n = len(offs) - 1
        p_ids, p_off = C.c_void_p(), C.c_void_p()
        tp = text.ctypes.data if len(text) else None
        self._check(self._lib.spmx_encode_batch(self._h, tp, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off)))
        try:
            io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(io[n])
            ids = (np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
        finally:
            self._lib.spmx_free(p_ids)
            self._lib.spmx_free(p_off)
        return ids, io

    def EncodeDevice(self, d_text, d_offsets, d_ids=None, d_id_offsets=None, stream=None,
                     add_bos=False, add_eos=False, reverse=False):
        """Device-resident form over torch tensors on this processor's GPU.

        ``d_text`` uint8[bytes], ``d_offsets`` int64/uint64[n + 1].  ``d_ids``
        (int32) and ``d_id_offsets`` (int64[n + 1]) are allocated when not
        given; a too-small ``d_ids`` is re-allocated once.  Returns
        ``(d_ids, d_id_offsets, total)``; ``d_ids[:total]`` is valid."""
        import torch
        self._need()
        self._apply(add_bos, add_eos, reverse)
        n = d_offsets.numel() - 1
        if d_id_offsets is None:
            d_id_offsets = torch.empty(n + 1, dtype=torch.int64, device=d_text.device)
        if d_ids is None:
            d_ids = torch.empty(d_text.numel() // 2 + 4 * n + 64, dtype=torch.int32, device=d_text.device)
        if stream is None:
            stream = torch.cuda.current_stream(d_text.device).cuda_stream if d_text.is_cuda else 0
        total = C.c_uint64(0)
        for _ in range(2):
            rc = self._lib.spmx_encode_batch_device(
                self._h, d_text.data_ptr(), d_text.numel(), d_offsets.data_ptr(), n, d_ids.data_ptr(),
                d_ids.numel(), d_id_offsets.data_ptr(), stream, C.byref(total))
            if rc == _RESOURCE_EXHAUSTED and total.value > d_ids.numel():
                d_ids = torch.empty(total.value, dtype=torch.int32, device=d_text.device)
                continue
            break
        self._check(rc)
        return d_ids, d_id_offsets, total.value

    # ------------------------------------------------------- spans form ----
    def EncodeSpansPacked(self, text, offsets, norm_spans=False):
        """Packed host arrays -> ``(ids int32, begin uint32, end uint32, id_offsets uint64)``: next to every id the
        byte range of its sentence it covers -- ``pieces[i].begin / .end`` of the ``SentencePieceText`` that
        ``Encode(input, SentencePieceText*)`` fills (src/sentencepiece_processor.cc:547-653), in bytes as in C++
        (the reference's Python wrapper converts to characters).  ``add_bos`` / ``add_eos`` / ``reverse`` are not
        taken here, as in the reference's proto API (sentencepiece.i:166-186); ``SetEncodeExtraOptions`` applies.
        ``norm_spans=True`` appends ``(nbegin, nend)``: the same tokens as ranges of the normalized sentence."""
        self._need()
        self._apply(False, False, False)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        p_ids, p_off, p_b, p_e, p_nb, p_ne = (C.c_void_p() for _ in range(6))
        tp = text.ctypes.data if len(text) else None
        self._check(self._lib.spmx_encode_batch_spans(self._h, tp, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off),
                                                      C.byref(p_b), C.byref(p_e),
                                                      C.byref(p_nb) if norm_spans else None,
                                                      C.byref(p_ne) if norm_spans else None))
        try:
            io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(io[n])

            def take(p, ct, dt):
                return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(total,)).copy() if total else np.zeros(0, dtype=dt)
            ids = take(p_ids, C.c_int32, np.int32)
            b = take(p_b, C.c_uint32, np.uint32)
            e = take(p_e, C.c_uint32, np.uint32)
            if norm_spans:
                nb = take(p_nb, C.c_uint32, np.uint32)
                ne = take(p_ne, C.c_uint32, np.uint32)
        finally:
            for p in (p_ids, p_off, p_b, p_e, p_nb, p_ne):
                self._lib.spmx_free(p)
        return (ids, b, e, io, nb, ne) if norm_spans else (ids, b, e, io)

    def EncodeSpansDevice(self, d_text, d_offsets, stream=None):
        """Device-resident spans form: ``(d_ids int32, d_id_offsets int64[n + 1], d_begin int32, d_end int32, total)``
        (the span tensors hold uint32 values; torch has no uint32 arithmetic, so they are typed int32)."""
        import torch
        self._need()
        self._apply(False, False, False)
        n = d_offsets.numel() - 1
        dev = d_text.device
        d_id_offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
        cap = d_text.numel() // 2 + 4 * n + 64
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        total = C.c_uint64(0)
        for _ in range(2):
            d_ids = torch.empty(cap, dtype=torch.int32, device=dev)
            d_b = torch.empty(cap, dtype=torch.int32, device=dev)
            d_e = torch.empty(cap, dtype=torch.int32, device=dev)
            rc = self._lib.spmx_encode_batch_spans_device(
                self._h, d_text.data_ptr(), d_text.numel(), d_offsets.data_ptr(), n, d_ids.data_ptr(), cap,
                d_id_offsets.data_ptr(), d_b.data_ptr(), d_e.data_ptr(), None, None, stream, C.byref(total))
            if rc == _RESOURCE_EXHAUSTED and total.value > cap:
                cap = total.value
                continue
            break
        self._check(rc)
        return d_ids, d_id_offsets, d_b, d_e, total.value

    def EncodeAsSentencePieceText(self, input):
        """str | bytes | list of them -> per sentence the list of ``(piece, id, surface, begin, end)`` -- the fields of
        ``SentencePieceText.pieces`` (src/sentencepiece.proto, PopulateSentencePieceText
        sentencepiece_processor.cc:547-636), all bytes.  ``piece`` is the token's normalized text (so an unknown
        token shows its characters, :614-617), the ``<0x..>`` name for a byte-fallback piece and the piece name for
        a bos / eos; the ``unk_piece`` extra option replaces the piece of unknown tokens (:1050-1058)."""
        single = isinstance(input, (str, bytes))
        items = [input] if single else list(input)
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in items]
        offs = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            np.cumsum([len(x) for x in raw], out=offs[1:])
        text = np.frombuffer(b"".join(raw), dtype=np.uint8)
        ids, b, e, io, nb, ne = self.EncodeSpansPacked(text, offs, norm_spans=True)
        norm, no, _ = self.NormalizePacked(text, offs)
        norm = norm.tobytes()
        unk = self.unk_id()
        unk_opt = any(o in ("unk", "unk_piece") for o in (self._extra or "").split(":"))
        out = []
        for i, r in enumerate(raw):
            base = int(no[i])
            row = []
            for k in range(int(io[i]), int(io[i + 1])):
                t = int(ids[k])
                if unk_opt and self.IsUnknown(t):             # model_->unk_piece() (:1050-1058), trainer_spec.unk_piece
                    piece = self.unk_piece().encode("utf-8")
                elif self.IsByte(t) or self.IsControl(t):
                    piece = self.IdToPiece(t).encode("utf-8")
                else:
                    piece = norm[base + int(nb[k]):base + int(ne[k])]
                row.append((piece, t, r[int(b[k]):int(e[k])], int(b[k]), int(e[k])))
            out.append(row)
        return out[0] if single else out

    def EncodeAsSerializedProto(self, input):
        """``EncodeAsSerializedProto`` (src/sentencepiece_processor.h:528-531) / ``encode(out_type="serialized_proto")``:
        the serialized ``SentencePieceText`` of every sentence (bytes; a list gives a list)."""
        from . import spt_proto
        single = isinstance(input, (str, bytes))
        items = [input] if single else list(input)
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in items]
        offs = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            np.cumsum([len(x) for x in raw], out=offs[1:])
        text = np.frombuffer(b"".join(raw), dtype=np.uint8)
        ids, b, e, io, nb, ne = self.EncodeSpansPacked(text, offs, norm_spans=True)
        rows = self.EncodeAsSentencePieceText(raw)
        rev = sum(1 for o in (self._extra or "").split(":") if o == "reverse") % 2 == 1
        out = []
        for i, r in enumerate(raw):
            lo, hi = int(io[i]), int(io[i + 1])
            has = spt_proto.surface_flags(ids[lo:hi], nb[lo:hi], self.IsByte, self.IsControl, rev)
            out.append(spt_proto.serialize(r, [(p, t, sf if has[k] else None, pb, pe)
                                               for k, (p, t, sf, pb, pe) in enumerate(rows[i])]))
        return out[0] if single else out

    def EncodeAsImmutableProto(self, input):
        """``encode(out_type="immutable_proto")``: per sentence an object with ``.text``, ``.pieces[i].piece / .id /
        .surface / .begin / .end`` and ``SerializeAsString()``.  As in the reference's Python wrapper, begin / end are
        converted from bytes to Unicode characters (``ConvertToUnicodeSpans``, src/sentencepiece_processor.cc:63-89);
        the serialized form keeps bytes."""
        from . import spt_proto
        single = isinstance(input, (str, bytes))
        items = [input] if single else list(input)
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in items]
        rows = self.EncodeAsSentencePieceText(raw)
        blobs = self.EncodeAsSerializedProto(raw)
        out = [spt_proto.ImmutableSentencePieceText(r, rows[i], blobs[i]) for i, r in enumerate(raw)]
        return out[0] if single else out

    def EncodeAsPieces(self, input):
        """``EncodeAsPieces`` (sentencepiece_processor.h:453-456) / ``encode(out_type=str)``: the piece strings."""
        single = isinstance(input, (str, bytes))
        rows = self.EncodeAsSentencePieceText([input] if single else input)
        out = [[p.decode("utf-8", "surrogateescape") for p, *_ in row] for row in rows]
        return out[0] if single else out

    encode_as_pieces = EncodeAsPieces
    encode_as_serialized_proto = EncodeAsSerializedProto
    encode_as_immutable_proto = EncodeAsImmutableProto

    # ------------------------------------------------------------ n-best ----
    def NBestPacked(self, text, offsets, nbest_size):
        """Packed host arrays -> ``(ids int32, id_offsets uint64[R + 1], scores float32[R], result_offsets uint64[n + 1])``:
        result r has ``ids[id_offsets[r]:id_offsets[r + 1]]`` and ``scores[r]``; sentence s owns results
        ``result_offsets[s]:result_offsets[s + 1]``, best first (``NBestEncode``, unigram models only)."""
        self._need()
        self._apply(False, False, False)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        p_ids, p_io, p_sc, p_ro = (C.c_void_p() for _ in range(4))
        self._check(self._lib.spmx_nbest_encode_batch(self._h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                                      int(nbest_size), C.byref(p_ids), C.byref(p_io), C.byref(p_sc),
                                                      C.byref(p_ro)))
        try:
            ro = np.ctypeslib.as_array(C.cast(p_ro, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            R = int(ro[n])
            io = np.ctypeslib.as_array(C.cast(p_io, C.POINTER(C.c_uint64)), shape=(R + 1,)).copy()
            total = int(io[R])
            ids = (np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
            sc = (np.ctypeslib.as_array(C.cast(p_sc, C.POINTER(C.c_float)), shape=(R,)).copy()
                  if R else np.zeros(0, dtype=np.float32))
        finally:
            for p in (p_ids, p_io, p_sc, p_ro):
                self._lib.spmx_free(p)
        return ids, io, sc, ro

    def NBestEncodeAsIds(self, input, nbest_size):
        """``NBestEncodeAsIds`` (python/src/sentencepiece/__init__.py ``nbest_encode_as_ids``): str -> list of id lists,
        best first; a list of str gives a list of those."""
        single = isinstance(input, (str, bytes))
        items = [input] if single else list(input)
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in items]
        offs = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            np.cumsum([len(x) for x in raw], out=offs[1:])
        ids, io, _, ro = self.NBestPacked(np.frombuffer(b"".join(raw), dtype=np.uint8), offs, nbest_size)
        out = [[ids[int(io[r]):int(io[r + 1])].tolist() for r in range(int(ro[s]), int(ro[s + 1]))] for s in range(len(raw))]
        return out[0] if single else out

    nbest_encode_as_ids = NBestEncodeAsIds

    def _spans_arrays(self, ptrs, total):
        out = []
        for p in ptrs:
            out.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(total,)).copy()
                       if total else np.zeros(0, dtype=np.uint32))
        return out

    def NBestSpansPacked(self, text, offsets, nbest_size):
        """As NBestPacked, plus per id of every result the byte range of the input (``begin`` / ``end``) and of the
        normalized text (``nbegin`` / ``nend``) its piece covers: ``(ids, id_offsets, scores, result_offsets, begin, end,
        nbegin, nend)`` (``NBestEncode(input, nbest_size, NBestSentencePieceText *)``,
        src/sentencepiece_processor.cc:653-676)."""
        self._need()
        self._apply(False, False, False)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        ps = [C.c_void_p() for _ in range(8)]
        self._check(self._lib.spmx_nbest_encode_batch_spans(self._h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                                            int(nbest_size), *[C.byref(p) for p in ps]))
        try:
            ro = np.ctypeslib.as_array(C.cast(ps[3], C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            R = int(ro[n])
            io = np.ctypeslib.as_array(C.cast(ps[1], C.POINTER(C.c_uint64)), shape=(R + 1,)).copy()
            total = int(io[R])
            ids = (np.ctypeslib.as_array(C.cast(ps[0], C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
            sc = (np.ctypeslib.as_array(C.cast(ps[2], C.POINTER(C.c_float)), shape=(R,)).copy()
                  if R else np.zeros(0, dtype=np.float32))
            b, e, nb, ne = self._spans_arrays(ps[4:], total)
        finally:
            for p in ps:
                self._lib.spmx_free(p)
        return ids, io, sc, ro, b, e, nb, ne

    def SampleSpansPacked(self, text, offsets, nbest_size, alpha, seed=0, _keep_options=False):
        """``SampleEncode(input, nbest_size, alpha, SentencePieceText *)`` per sentence: ``(ids, id_offsets, begin, end,
        nbegin, nend)`` of the drawn segmentations."""
        self._need()
        if not _keep_options:         # (Encode(out_type=str, enable_sampling=True) has compiled add_bos / add_eos / reverse in)
            self._apply(False, False, False)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        ps = [C.c_void_p() for _ in range(6)]
        self._check(self._lib.spmx_sample_encode_batch_spans(self._h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                                             int(nbest_size), C.c_float(alpha), int(seed), *[C.byref(p) for p in ps]))
        try:
            io = np.ctypeslib.as_array(C.cast(ps[1], C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(io[n])
            ids = (np.ctypeslib.as_array(C.cast(ps[0], C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
            b, e, nb, ne = self._spans_arrays(ps[2:], total)
        finally:
            for p in ps:
                self._lib.spmx_free(p)
        return ids, io, b, e, nb, ne

    def _rows_from_spans(self, raw, norm, no, sent_of_row, ids, io, b, e, nb, ne):
        """Rows of ``(piece, id, surface | None, begin, end)`` (PopulateSentencePieceText's fields, all bytes) for the CSR
        rows ``io``; row r belongs to sentence ``sent_of_row[r]``.  surface None: a field the reference leaves unset."""
        from . import spt_proto
        unk_opt = any(o in ("unk", "unk_piece") for o in (self._applied or "").split(":"))
        rev = sum(1 for o in (self._applied or "").split(":") if o == "reverse") % 2 == 1
        out = []
        for r in range(len(io) - 1):
            i = sent_of_row[r]
            base = int(no[i])
            lo, hi = int(io[r]), int(io[r + 1])
            has = spt_proto.surface_flags(ids[lo:hi], nb[lo:hi], self.IsByte, self.IsControl, rev)
            row = []
            for k in range(lo, hi):
                t = int(ids[k])
                if unk_opt and self.IsUnknown(t):
                    piece = self.unk_piece().encode("utf-8")
                elif self.IsByte(t) or self.IsControl(t):
                    piece = self.IdToPiece(t).encode("utf-8")
                else:
                    piece = norm[base + int(nb[k]):base + int(ne[k])]
                row.append((piece, t, raw[i][int(b[k]):int(e[k])] if has[k - lo] else None, int(b[k]), int(e[k])))
            out.append(row)
        return out

    def _pack_items(self, input):
        single = isinstance(input, (str, bytes))
        items = [input] if single else list(input)
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in items]
        offs = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            np.cumsum([len(x) for x in raw], out=offs[1:])
        return single, raw, np.frombuffer(b"".join(raw), dtype=np.uint8), offs

    def NBestEncodeAsSentencePieceText(self, input, nbest_size):
        """Per sentence the list of its results, best first, each ``(score, rows)`` with rows of ``(piece, id, surface |
        None, begin, end)`` (bytes): the fields of ``NBestSentencePieceText`` (src/sentencepiece.proto)."""
        single, raw, text, offs = self._pack_items(input)
        ids, io, sc, ro, b, e, nb, ne = self.NBestSpansPacked(text, offs, nbest_size)
        norm, no, _ = self.NormalizePacked(text, offs)
        sent = [s for s in range(len(raw)) for _ in range(int(ro[s]), int(ro[s + 1]))]
        rows = self._rows_from_spans(raw, norm.tobytes(), no, sent, ids, io, b, e, nb, ne)
        out = [[(float(sc[r]), rows[r]) for r in range(int(ro[s]), int(ro[s + 1]))] for s in range(len(raw))]
        return out[0] if single else out

    def NBestEncodeAsPieces(self, input, nbest_size):
        """``NBestEncodeAsPieces`` (src/sentencepiece_processor.h:318-320; python ``nbest_encode_as_pieces``)."""
        single = isinstance(input, (str, bytes))
        res = self.NBestEncodeAsSentencePieceText([input] if single else input, nbest_size)
        out = [[[p.decode("utf-8", "surrogateescape") for p, *_ in rows] for _, rows in per] for per in res]
        return out[0] if single else out

    def NBestEncodeAsSerializedProto(self, input, nbest_size):
        """``NBestEncodeAsSerializedProto`` (src/sentencepiece_processor.h:545-550): the serialized
        ``NBestSentencePieceText`` of every sentence (each result carries its score)."""
        from . import spt_proto
        single, raw, _, _ = self._pack_items(input)
        res = self.NBestEncodeAsSentencePieceText(raw, nbest_size)
        out = [spt_proto.serialize_nbest([spt_proto.serialize(raw[i], rows, score) for score, rows in per])
               for i, per in enumerate(res)]
        return out[0] if single else out

    nbest_encode_as_pieces = NBestEncodeAsPieces
    nbest_encode_as_serialized_proto = NBestEncodeAsSerializedProto

    def SampleEncodeAsSentencePieceText(self, input, nbest_size, alpha, seed=None, _keep_options=False):
        """The drawn segmentation of every sentence as rows of ``(piece, id, surface | None, begin, end)``
        (``SampleEncode(input, nbest_size, alpha, SentencePieceText *)``, src/sentencepiece_processor.cc:678-720)."""
        if seed is None:
            if not hasattr(self, "_sample_calls"):
                self._sample_calls = int.from_bytes(os.urandom(7), "little")
            self._sample_calls += 1
            seed = self._sample_calls
        single, raw, text, offs = self._pack_items(input)
        ids, io, b, e, nb, ne = self.SampleSpansPacked(text, offs, nbest_size, alpha, seed, _keep_options=_keep_options)
        norm, no, _ = self.NormalizePacked(text, offs)
        rows = self._rows_from_spans(raw, norm.tobytes(), no, list(range(len(raw))), ids, io, b, e, nb, ne)
        return rows[0] if single else rows

    def SampleEncodeAsPieces(self, input, nbest_size, alpha, seed=None, _keep_options=False):
        """``SampleEncodeAsPieces`` (src/sentencepiece_processor.h:404-408; python ``sample_encode_as_pieces``)."""
        single = isinstance(input, (str, bytes))
        rows = self.SampleEncodeAsSentencePieceText([input] if single else input, nbest_size, alpha, seed, _keep_options=_keep_options)
        out = [[p.decode("utf-8", "surrogateescape") for p, *_ in row] for row in rows]
        return out[0] if single else out

    def SampleEncodeAsSerializedProto(self, input, nbest_size, alpha, seed=None):
        """``SampleEncodeAsSerializedProto`` (src/sentencepiece_processor.h:539-543)."""
        from . import spt_proto
        single, raw, _, _ = self._pack_items(input)
        rows = self.SampleEncodeAsSentencePieceText(raw, nbest_size, alpha, seed)
        out = [spt_proto.serialize(raw[i], rows[i]) for i in range(len(raw))]
        return out[0] if single else out

    sample_encode_as_pieces = SampleEncodeAsPieces
    sample_encode_as_serialized_proto = SampleEncodeAsSerializedProto

    # ---- the remaining spellings of the reference's Python wrapper for this path (python/src/sentencepiece/__init__.py) ----
    def SampleEncodeAsIds(self, input, nbest_size=None, alpha=None, **kwargs):
        """``SampleEncodeAsIds`` (:586-588)."""
        return self.Encode(input, nbest_size=nbest_size, alpha=alpha, out_type=int, enable_sampling=True, **kwargs)

    def SampleEncodeAsImmutableProto(self, input, nbest_size=None, alpha=None, **kwargs):
        """``SampleEncodeAsImmutableProto`` (:596-598)."""
        return self.Encode(input, nbest_size=nbest_size, alpha=alpha, out_type="immutable_proto", enable_sampling=True, **kwargs)

    sample_encode_as_ids = SampleEncodeAsIds
    sample_encode_as_immutable_proto = SampleEncodeAsImmutableProto

    def NBestEncode(self, input, out_type=None, add_bos=None, add_eos=None, reverse=None, emit_unk_piece=None, nbest_size=None):
        """``NBestEncode`` (:601-656): out_type int | str | "serialized_proto" | "immutable_proto"; add_bos / add_eos /
        reverse / emit_unk_piece rewrite the id and piece forms (RewriteIds, sentencepiece.i:138-164) and are refused by
        the proto forms (:177-186)."""
        from . import spt_proto
        self._need()
        out_type = self._out_type if out_type is None else out_type
        a_bos = self._add_bos if add_bos is None else add_bos
        a_eos = self._add_eos if add_eos is None else add_eos
        rev = self._reverse if reverse is None else reverse
        emit = self._emit_unk_piece if emit_unk_piece is None else emit_unk_piece
        nb = self._nbest_size if nbest_size is None else nbest_size
        if nb <= 0:
            nb = 1
        single, raw, text, offs = self._pack_items(input)
        if out_type is int or out_type is str or out_type == "str":
            self._apply(a_bos, a_eos, rev)
            try:
                ids, io, sc, ro, b, e, nbg, nen = self._nbest_spans_keep(text, offs, nb)
                if out_type is int:
                    out = [[ids[int(io[r]):int(io[r + 1])].tolist() for r in range(int(ro[s]), int(ro[s + 1]))] for s in range(len(raw))]
                else:
                    norm, no, _ = self.NormalizePacked(text, offs)
                    sent = [s for s in range(len(raw)) for _ in range(int(ro[s]), int(ro[s + 1]))]
                    rows = self._rows_from_spans(raw, norm.tobytes(), no, sent, ids, io, b, e, nbg, nen)
                    unk = self.IdToPiece(self.unk_id())
                    out = [[[(unk if emit and self.IsUnknown(t) else p.decode("utf-8", "surrogateescape")) for p, t, *_ in rows[r]]
                            for r in range(int(ro[s]), int(ro[s + 1]))] for s in range(len(raw))]
            finally:
                self._apply(False, False, False)
            return out[0] if single else out
        if any([a_bos, a_eos, rev, emit]):
            raise NotImplementedError("add_bos, add_eos, reverse, and emit_unk_piece is not supported in proto API")
        if out_type in ("serialized_proto", "proto"):
            return self.NBestEncodeAsSerializedProto(input, nb)
        if out_type == "immutable_proto":
            res = self.NBestEncodeAsSentencePieceText(raw, nb)
            out = []
            for i, per in enumerate(res):
                views = []
                for score, rows in per:
                    v = spt_proto.ImmutableSentencePieceText(raw[i], [(p, t, sf if sf is not None else b"", pb, pe) for p, t, sf, pb, pe in rows],
                                                             spt_proto.serialize(raw[i], rows, score))
                    v.score = score
                    views.append(v)
                out.append(spt_proto.ImmutableNBestSentencePieceText(views))
            return out[0] if single else out
        raise RuntimeError("unknown out_type")

    def _nbest_spans_keep(self, text, offs, nbest_size):
        """NBestSpansPacked without resetting the options compiled into the handle (NBestEncode has applied its own)."""
        n = len(offs) - 1
        ps = [C.c_void_p() for _ in range(8)]
        self._check(self._lib.spmx_nbest_encode_batch_spans(self._h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                                            int(nbest_size), *[C.byref(p) for p in ps]))
        try:
            ro = np.ctypeslib.as_array(C.cast(ps[3], C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            R = int(ro[n])
            io = np.ctypeslib.as_array(C.cast(ps[1], C.POINTER(C.c_uint64)), shape=(R + 1,)).copy()
            total = int(io[R])
            ids = (np.ctypeslib.as_array(C.cast(ps[0], C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
            sc = (np.ctypeslib.as_array(C.cast(ps[2], C.POINTER(C.c_float)), shape=(R,)).copy()
                  if R else np.zeros(0, dtype=np.float32))
            b, e, nb, ne = self._spans_arrays(ps[4:], total)
        finally:
            for p in ps:
                self._lib.spmx_free(p)
        return ids, io, sc, ro, b, e, nb, ne

    def NBestEncodeAsImmutableProto(self, input, nbest_size=None, **kwargs):
        """``NBestEncodeAsImmutableProto`` (:674-676): ``.nbests[i]`` with ``.score``, ``.text``, ``.pieces``."""
        return self.NBestEncode(input, nbest_size=nbest_size, out_type="immutable_proto", **kwargs)

    nbest_encode = NBestEncode
    nbest_encode_as_immutable_proto = NBestEncodeAsImmutableProto

    def Tokenize(self, input, **kwargs):
        """``Tokenize`` = ``Encode`` (:1009-1010)."""
        return self.Encode(input, **kwargs)

    def Detokenize(self, input, **kwargs):
        """``Detokenize`` = ``Decode`` (:1013-1014)."""
        return self.Decode(input, **kwargs)

    tokenize, detokenize = Tokenize, Detokenize

    def DecodePieces(self, input, out_type=str, **kwargs):
        """``DecodePieces`` (:871-872, ``_DecodePiecesBatch`` sentencepiece.i:547): pieces -> text.  A piece that is not in
        the vocabulary is copied through as text (src/sentencepiece_processor.cc:784-790): it travels to the decode
        kernels as a literal beside the ids (``spmx_decode_batch_pieces``); under a decode extra option ``unk`` the
        reference rewrites it to the unknown piece first (.cc:1050-1058)."""
        self._need()
        single = not input or isinstance(input[0], (str, bytes))
        items = [input] if single else input
        unk = self.unk_id()
        unk_name = self.IdToPiece(unk).encode("utf-8") if unk >= 0 else None
        to_unk = bool(self._lib.spmx_decode_unk_option(self._h))
        ids, lits = [], []
        offs = np.zeros(len(items) + 1, dtype=np.uint64)
        for r, row in enumerate(items):
            for p in row:
                pb = p if isinstance(p, bytes) else p.encode("utf-8", "surrogateescape")
                t = self._lib.spmx_piece_to_id(self._h, pb, len(pb))
                if t == unk and pb != unk_name and not to_unk:
                    lits.append(pb)
                    t = -len(lits)
                ids.append(t)
            offs[r + 1] = len(ids)
        ids = np.asarray(ids, dtype=np.int32)
        lo = np.zeros(len(lits) + 1, dtype=np.uint64)
        if lits:
            np.cumsum([len(x) for x in lits], out=lo[1:])
        blob = b"".join(lits)
        p_text, p_off = C.c_void_p(), C.c_void_p()
        self._check(self._lib.spmx_decode_batch_pieces(self._h, ids.ctypes.data if len(ids) else None, offs.ctypes.data, len(items),
                                                       blob if blob else None, lo.ctypes.data, len(lits), C.byref(p_text), C.byref(p_off)))
        try:
            to = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(len(items) + 1,)).astype(np.int64)
            total = int(to[-1])
            b = bytes(np.ctypeslib.as_array(C.cast(p_text, C.POINTER(C.c_uint8)), shape=(total,))) if total else b""
        finally:
            self._lib.spmx_free(p_text)
            self._lib.spmx_free(p_off)
        out = [b[to[i]:to[i + 1]] for i in range(len(items))]
        if out_type is str:
            out = [x.decode("utf-8", errors="replace") for x in out]
        return out[0] if single else out

    decode_pieces = DecodePieces

    def GetScore(self, id):
        """``GetScore`` (:285-286): the piece's score from the ModelProto."""
        self._need()
        if not hasattr(self, "_scores") or self._scores[0] is not self._model_proto:
            from . import model_proto_scores
            self._scores = (self._model_proto, model_proto_scores.scores(self._model_proto))
        if id < 0 or id >= len(self._scores[1]):
            raise IndexError("piece id is out of range.")
        return self._scores[1][id]

    get_score = GetScore
    get_piece_size = lambda self: self.GetPieceSize()  # noqa: E731

    # -------------------------------------------------------- normalize ----
    def NormalizePacked(self, text, offsets, with_offsets=False):
        """Packed host arrays -> ``(normalized uint8, norm_offsets uint64[n + 1], norm_to_orig | None)``.
        ``norm_to_orig`` (uint32): sentence s owns ``[norm_offsets[s] + s, norm_offsets[s + 1] + s + 1)`` -- one entry
        per normalized byte plus the closing one (0xFFFFFFFF where the reference's vector is empty)."""
        self._need()
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        p_t, p_o, p_a = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._lib.spmx_normalize_batch(self._h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                                   C.byref(p_t), C.byref(p_o), C.byref(p_a) if with_offsets else None))
        try:
            no = np.ctypeslib.as_array(C.cast(p_o, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(no[n])
            norm = (np.ctypeslib.as_array(C.cast(p_t, C.POINTER(C.c_uint8)), shape=(total,)).copy()
                    if total else np.zeros(0, dtype=np.uint8))
            n2o = None
            if with_offsets:
                n2o = (np.ctypeslib.as_array(C.cast(p_a, C.POINTER(C.c_uint32)), shape=(total + n,)).copy()
                       if total + n else np.zeros(0, dtype=np.uint32))
        finally:
            for p in (p_t, p_o, p_a):
                self._lib.spmx_free(p)
        return norm, no, n2o

    def Normalize(self, input, with_offsets=None):
        """``Normalize`` (python/src/sentencepiece/__init__.py:907-915): str -> str, or ``(str, list[int])`` with
        ``with_offsets`` (byte offsets, the C++ norm_to_orig); a list gives a list."""
        single = not isinstance(input, list)
        items = [input] if single else input
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in items]
        offs = np.zeros(len(raw) + 1, dtype=np.uint64)
        if raw:
            np.cumsum([len(x) for x in raw], out=offs[1:])
        norm, no, n2o = self.NormalizePacked(np.frombuffer(b"".join(raw), dtype=np.uint8), offs, bool(with_offsets))
        nb = norm.tobytes()
        out = []
        for i in range(len(raw)):
            s = nb[int(no[i]):int(no[i + 1])].decode("utf-8", "surrogateescape")
            if with_offsets:
                a = n2o[int(no[i]) + i:int(no[i + 1]) + i + 1]
                out.append((s, [] if (len(a) == 1 and a[0] == 0xFFFFFFFF) else [int(v) for v in a]))
            else:
                out.append(s)
        return out[0] if single else out

    normalize = Normalize

    # ----------------------------------------------------------- decode ----
    def Decode(self, input, out_type=str, num_threads=None):
        """list[int] -> str; list[list[int]] -> list[str] (``Decode`` / ``_DecodeIdsBatch``,
        python/src/sentencepiece/__init__.py:808-870).  ``out_type=bytes`` returns the raw bytes."""
        self._need()
        single = not input or isinstance(input[0], (int, np.integer))
        items = [input] if single else input
        offs = np.zeros(len(items) + 1, dtype=np.uint64)
        if items:
            np.cumsum([len(x) for x in items], out=offs[1:])
        ids = np.fromiter((t for x in items for t in x), dtype=np.int32, count=int(offs[-1]))
        text, to = self.DecodePacked(ids, offs)
        b = text.tobytes()
        to = to.astype(np.int64)
        out = [b[to[i]:to[i + 1]] for i in range(len(items))]
        if out_type is str:
            out = [x.decode("utf-8", errors="replace") for x in out]
        return out[0] if single else out

    decode = DecodeIds = decode_ids = Decode

    def DecodePacked(self, ids, id_offsets):
        """CSR host arrays ``(ids int32, id_offsets uint64[n + 1])`` -> packed ``(text uint8, text_offsets uint64)``."""
        self._need()
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        id_offsets = np.ascontiguousarray(id_offsets, dtype=np.uint64)
        n = len(id_offsets) - 1
        p_text, p_off = C.c_void_p(), C.c_void_p()
        ip = ids.ctypes.data if len(ids) else None
        self._check(self._lib.spmx_decode_batch(self._h, ip, id_offsets.ctypes.data, n, C.byref(p_text), C.byref(p_off)))
        try:
            to = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(to[n])
            text = (np.ctypeslib.as_array(C.cast(p_text, C.POINTER(C.c_uint8)), shape=(total,)).copy()
                    if total else np.zeros(0, dtype=np.uint8))
        finally:
            self._lib.spmx_free(p_text)
            self._lib.spmx_free(p_off)
        return text, to

    def DecodeDevice(self, d_ids, d_id_offsets, d_text=None, d_text_offsets=None, stream=None):
        """Device-resident form over torch tensors: ``d_ids`` int32, ``d_id_offsets`` int64[n + 1] ->
        ``(d_text uint8, d_text_offsets int64[n + 1], total_bytes)``; ``d_text[:total_bytes]`` is valid."""
        import torch
        self._need()
        n = d_id_offsets.numel() - 1
        if d_text_offsets is None:
            d_text_offsets = torch.empty(n + 1, dtype=torch.int64, device=d_ids.device)
        if d_text is None:
            d_text = torch.empty(d_ids.numel() * 6 + 64, dtype=torch.uint8, device=d_ids.device)
        if stream is None:
            stream = torch.cuda.current_stream(d_ids.device).cuda_stream
        total = C.c_uint64(0)
        for _ in range(2):
            rc = self._lib.spmx_decode_batch_device(self._h, d_ids.data_ptr(), d_id_offsets.data_ptr(), n,
                                                    d_text.data_ptr(), d_text.numel(), d_text_offsets.data_ptr(),
                                                    stream, C.byref(total))
            if rc == _RESOURCE_EXHAUSTED and total.value > d_text.numel():
                d_text = torch.empty(total.value, dtype=torch.uint8, device=d_ids.device)
                continue
            break
        self._check(rc)
        return d_text, d_text_offsets, total.value

    # ---------------------------------------------------- corpus packer ----
    def SplitLinesDevice(self, d_file, stream=None):
        """A file image on the GPU (torch uint8 tensor, '\\n'-terminated lines, std::getline semantics) ->
        ``(d_text uint8, d_offsets int64[n + 1], n_lines)``: the packed form ``EncodeDevice`` takes."""
        import torch
        self._need()
        nbytes = d_file.numel()
        if stream is None:
            stream = torch.cuda.current_stream(d_file.device).cuda_stream
        n, tb = C.c_uint64(0), C.c_uint64(0)
        d_text = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=d_file.device)
        d_offs = torch.empty(nbytes // 24 + 1024, dtype=torch.int64, device=d_file.device)
        for _ in range(2):
            rc = self._lib.spmx_split_lines_device(self._h, d_file.data_ptr(), nbytes, d_text.data_ptr(), d_text.numel(),
                                                   d_offs.data_ptr(), d_offs.numel(), stream, C.byref(n), C.byref(tb))
            if rc == _RESOURCE_EXHAUSTED:
                d_offs = torch.empty(n.value + 1, dtype=torch.int64, device=d_file.device)
                continue
            break
        self._check(rc)
        return d_text[:tb.value], d_offs[:n.value + 1], n.value

    # ------------------------------------------------------ measurement ----
    def _csr_call(self, fn, text, offsets, *mid):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        p_ids, p_off = C.c_void_p(), C.c_void_p()
        self._check(fn(self._h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, *mid, C.byref(p_ids), C.byref(p_off)))
        try:
            io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            total = int(io[n])
            ids = (np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(total,)).copy()
                   if total else np.zeros(0, dtype=np.int32))
        finally:
            self._lib.spmx_free(p_ids)
            self._lib.spmx_free(p_off)
        return ids, io

    def SampleEncodePacked(self, text, offsets, nbest_size=-1, alpha=0.1, seed=0):
        """SampleEncode per sentence (subword regularization; src/sentencepiece_processor.cc:678-720): packed host arrays
        -> CSR ``(ids, id_offsets)``.  ``nbest_size < 0`` samples from the whole lattice (unigram) / applies BPE-dropout
        with probability ``alpha`` (BPE); ``nbest_size > 1`` draws one of the n best.  ``seed`` keys the generators."""
        self._need()
        self._apply(False, False, False)
        return self._csr_call(self._lib.spmx_sample_encode_batch, text, offsets, int(nbest_size), C.c_float(alpha), int(seed))

    def SampleEncodeAsIds(self, input, nbest_size=-1, alpha=0.1, seed=0):
        single = not isinstance(input, list)
        items = [input] if single else input
        bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in items]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            np.cumsum([len(b) for b in bs], out=offs[1:])
        ids, io = self.SampleEncodePacked(np.frombuffer(b"".join(bs), dtype=np.uint8), offs, nbest_size, alpha, seed)
        io = io.astype(np.int64)
        out = [ids[io[i]:io[i + 1]].tolist() for i in range(len(bs))]
        return out[0] if single else out

    def EncodeOriginalPacked(self, text, offsets):
        """The reference's kOriginal unigram encoder (Lattice::Viterbi, src/unigram_model.cc:161-198, :674-692) per
        sentence -> CSR ``(ids, id_offsets)``."""
        self._need()
        self._apply(False, False, False)
        return self._csr_call(self._lib.spmx_encode_batch_original, text, offsets)

    def EncodeFile(self, in_path, out_path, output_format="id"):
        """Corpus file -> ids, the loop of the reference's ``spm_encode --output_format=id`` (spm_encode_main.cc:115-165)
        as one pipelined call: ``output_format="id"`` writes the reference's text (one line of space-separated ids per
        input line), ``"bin"`` the flat int32 ids to ``out_path`` and the uint64 offsets to ``out_path + ".idx"``.
        ``SetEncodeExtraOptions`` applies.  Returns ``(sentences, ids)``."""
        self._need()
        self._apply(False, False, False)
        ns, ni = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.spmx_encode_file(self._h, os.fsencode(in_path), os.fsencode(out_path),
                                               output_format.encode(), C.byref(ns), C.byref(ni)))
        return int(ns.value), int(ni.value)

    def SetProfiling(self, enabled):
        self._need()
        self._lib.spmx_set_profiling(self._h, 1 if enabled else 0)

    def HandleInfo(self):
        """dict(table_bytes, load_ms): what loading this handle cost (include/spmx.h spmx_handle_info; no reference counterpart)."""
        tb, ms = C.c_uint64(0), C.c_double(0.0)
        self._check(self._lib.spmx_handle_info(self._h, C.byref(tb), C.byref(ms)))
        return {"table_bytes": int(tb.value), "load_ms": float(ms.value)}

    def LastProfile(self):
        """Per kernel slot of the last profiled encode call (0 main streaming launch, 1 document launch, 2 overflow
        launch, 3 sentence-per-wave BPE, 4 long / wave-cooperative form, 5 word form first round, 6 second round): dict(kernel, kernel_ms, sentences, raw_bytes, ids, bytes,
        phase_cycles) + total_ms + path (sentences that waited in a wave's backlog / went to the overflow list / took
        the long form / failed)."""
        self._need()
        ms = np.zeros(8, dtype=np.float32)
        sent, raw, ids, byt = (np.zeros(8, dtype=np.uint64) for _ in range(4))
        path = np.zeros(8, dtype=np.uint64)
        tot = C.c_float(0)
        k = self._lib.spmx_last_profile(self._h, ms.ctypes.data, sent.ctypes.data, raw.ctypes.data, ids.ctypes.data,
                                        byt.ctypes.data, path.ctypes.data, C.byref(tot))
        cyc = np.zeros(80, dtype=np.uint64)
        self._lib.spmx_last_phase_cycles(self._h, cyc.ctypes.data)
        names = []
        for c in range(k):
            buf = C.create_string_buffer(64)
            self._lib.spmx_last_profile_name(self._h, c, buf, 64)
            names.append(buf.value.decode())
        return dict(classes=[dict(kernel=names[c], kernel_ms=float(ms[c]), sentences=int(sent[c]), raw_bytes=int(raw[c]),
                                  ids=int(ids[c]), bytes=int(byt[c]),
                                  phase_cycles=dict(zip(("load", "normalize", "segment", "emit", "search_trips"),
                                                        (int(x) for x in cyc[5 * c:5 * c + 5]))))
                             for c in range(k)],
                    path=dict(backlog=int(path[0]), overflow=int(path[1]), long=int(path[2]), failed=int(path[3])),
                    total_ms=float(tot.value))
