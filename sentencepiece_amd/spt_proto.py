"""Wire format of the reference's ``SentencePieceText`` message (src/sentencepiece.proto): what
``EncodeAsSerializedProto`` (src/sentencepiece_processor.h:528-531) returns.  Host-side assembly only: the fields come
from the device (ids, byte ranges, normalized-text ranges, normalized text); no protobuf runtime is needed.

  message SentencePieceText {
    optional string text = 1;
    message SentencePiece { optional string piece = 1; optional uint32 id = 2; optional string surface = 3;
                            optional uint32 begin = 4; optional uint32 end = 5; }
    repeated SentencePiece pieces = 2;
    optional float score = 3;          // not set by Encode
  }

proto2 presence: a field is written iff it was set.  PopulateSentencePieceText (src/sentencepiece_processor.cc:547-636)
sets piece / id / begin / end on every piece and surface on every piece except the byte-fallback pieces before the last
one of their character (:598-603); the bos / eos pieces of the extra options carry no surface (:1029-1048)."""


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _bytes_field(num, b):
    return _varint(num << 3 | 2) + _varint(len(b)) + b


def _uint_field(num, v):
    return _varint(num << 3) + _varint(v)


def serialize(text, pieces, score=None):
    """``text`` bytes; ``pieces`` iterable of ``(piece bytes, id, surface bytes | None, begin, end)``; ``score``: the
    float NBestEncode sets on every result (src/sentencepiece_processor.cc:670), None where the reference leaves the
    field unset (Encode, SampleEncode)."""
    import struct
    out = bytearray(_bytes_field(1, text))
    for piece, pid, surface, begin, end in pieces:
        m = _bytes_field(1, piece) + _uint_field(2, pid)
        if surface is not None:
            m += _bytes_field(3, surface)
        m += _uint_field(4, begin) + _uint_field(5, end)
        out += _bytes_field(2, m)
    if score is not None:
        out += _varint(3 << 3 | 5) + struct.pack("<f", score)
    return bytes(out)


def serialize_nbest(blobs):
    """``NBestSentencePieceText { repeated SentencePieceText nbests = 1; }`` from the serialized results."""
    return b"".join(_bytes_field(1, b) for b in blobs)


def surface_flags(ids, nbegin, is_byte, is_control, reversed_order):
    """Which pieces of one sentence carry a surface.  ``ids`` / ``nbegin`` in output order; ``is_byte(id)`` /
    ``is_control(id)`` piece-type predicates; ``reversed_order``: the ``reverse`` extra option is in effect.
    The byte pieces of one unknown character are consecutive and share their normalized begin; only the one that
    was last in text order has the surface."""
    n = len(ids)
    has = [True] * n
    for k in range(n):
        t = int(ids[k])
        if is_control(t):
            has[k] = False
        elif is_byte(t):
            nxt = k - 1 if reversed_order else k + 1       # the piece that follows in text order
            if 0 <= nxt < n and is_byte(int(ids[nxt])) and int(nbegin[nxt]) == int(nbegin[k]):
                has[k] = False
    return has


_CHAR_LEN = (1,) * 12 + (2, 2, 3, 4)      # string_util::OneCharLen (src/util.h:151-153) by the lead byte's high nibble


def utf8_to_unicode(text):
    """ConvertToUnicodeSpansInternal's table (src/sentencepiece_processor.cc:66-79): byte offset -> character index."""
    n = len(text)
    tab = [0] * (n + 1)
    prev = ulen = 0
    while prev < n:
        mb = _CHAR_LEN[text[prev] >> 4]
        for i in range(prev, min(prev + mb, n + 1)):
            tab[i] = ulen
        ulen += 1
        prev += mb
    tab[min(prev, n)] = ulen
    return tab


class ImmutableSentencePiece:
    __slots__ = ("piece", "id", "surface", "begin", "end")

    def __init__(self, piece, pid, surface, begin, end):
        self.piece, self.id, self.surface, self.begin, self.end = piece, pid, surface, begin, end

    def __repr__(self):
        return "piece: %r\nid: %d\nsurface: %r\nbegin: %d\nend: %d\n" % (self.piece, self.id, self.surface, self.begin, self.end)


class ImmutableSentencePieceText:
    """The read-only view the reference's Python wrapper returns for ``out_type="immutable_proto"``: strings decoded,
    spans in characters."""

    def __init__(self, text, rows, serialized):
        tab = utf8_to_unicode(text) if text else [0]
        hi = len(tab) - 1
        self.text = text.decode("utf-8", "surrogateescape")
        self.pieces = [ImmutableSentencePiece(p.decode("utf-8", "surrogateescape"), t, sf.decode("utf-8", "surrogateescape"),
                                              tab[min(max(b, 0), hi)] if text else b, tab[min(max(e, 0), hi)] if text else e)
                       for p, t, sf, b, e in rows]
        self.score = 0.0
        self._blob = serialized

    def SerializeAsString(self):
        return self._blob

    def __len__(self):
        return len(self.pieces)

    def __iter__(self):
        return iter(self.pieces)


class ImmutableNBestSentencePieceText:
    """The read-only view of ``NBestSentencePieceText``: ``.nbests`` (each with ``.score``) and ``SerializeAsString()``."""

    def __init__(self, views):
        self.nbests = list(views)

    def SerializeAsString(self):
        return serialize_nbest([v.SerializeAsString() for v in self.nbests])

    def __len__(self):
        return len(self.nbests)

    def __iter__(self):
        return iter(self.nbests)
