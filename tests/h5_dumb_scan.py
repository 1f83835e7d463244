"""A second, deliberately dumb look at an HDF5 file - test infrastructure that pins ``artist_amd.h5lite``.

h5py is not installed in this image, so the scenario-file fixtures were generated with h5lite on BOTH sides (it stands in
for h5py inside the reference and it is the product's reader): a reader bug would cancel.  This module shares no code and
no traversal with h5lite.  It never walks the group hierarchy by name: it greps the raw bytes for the signatures every old-style
group leaves behind - ``SNOD`` (symbol-table nodes: the links), ``TREE`` (the B-tree nodes that list them) and ``HEAP``
(local heaps: the link names) -, learns which heap belongs to which B-tree from the links that point at groups (their
scratch-pad when the pair is cached there, else message 0x0011 of the linked header; the root's pair sits in the superblock),
and decodes just enough of each linked version-1 object header (dataspace 0x0001, datatype 0x0003, data
layout 0x0008) to locate a dataset's raw bytes.  Result: ``{(leaf name, sha256 of the raw bytes, shape, item size)}``.

Format facts used (HDF5 File Format Specification 3.0, sections III.B, III.D, IV.A.1, IV.A.2): 8-byte offsets and
lengths; a symbol-table entry is 40 bytes (name offset, header address, cache type, reserved, 16 bytes scratch); a
version-1 object header is 16 bytes of prefix + 8-byte aligned messages (type u16, size u16, flags u8, 3 reserved);
message 0x0010 continues the header elsewhere.
"""
from __future__ import annotations

import hashlib
import struct


def _u(fmt, buf, off):
    return struct.unpack_from("<" + fmt, buf, off)


def _messages(buf, addr):
    """(type, body) of every message of the version-1 object header at ``addr`` (continuation blocks followed)."""
    version, _, n_msgs, _, size = _u("BBHII", buf, addr)
    if version != 1:
        return []
    blocks, out = [(addr + 16, size)], []
    while blocks and len(out) < n_msgs:
        off, length = blocks.pop(0)
        end = off + length
        while off + 8 <= end and len(out) < n_msgs:
            mtype, msize, _flags = _u("HHB", buf, off)
            body = buf[off + 8: off + 8 + msize]
            if mtype == 0x0010:
                blocks.append(_u("QQ", body, 0))
            out.append((mtype, body))
            off += 8 + msize
    return out


def _dataset(buf, addr):
    """(shape, item size, raw bytes) of the dataset whose object header is at ``addr``; None if it is not a contiguous or
    compact dataset of fixed-size items."""
    shape = item = raw = None
    for mtype, body in _messages(buf, addr):
        if mtype == 0x0001:                                    # dataspace
            version, rank, flags = _u("BBB", body, 0)
            dims_at = 8 if version == 1 else 4
            shape = tuple(_u("Q" * rank, body, dims_at)) if rank else ()
        elif mtype == 0x0003:                                  # datatype: class in the low nibble, size at byte 4
            klass = body[0] & 0x0F
            item = _u("I", body, 4)[0]
            if klass == 9:                                     # variable length: data lives in a global heap
                item = None
        elif mtype == 0x0008:                                  # data layout, version 3
            if body[0] != 3:
                return None
            if body[1] == 1:                                   # contiguous: address, size
                at, size = _u("QQ", body, 2)
                raw = b"" if at == 0xFFFFFFFFFFFFFFFF else bytes(buf[at: at + size])
            elif body[1] == 0:                                 # compact: size u16, data inline
                size = _u("H", body, 2)[0]
                raw = bytes(body[4: 4 + size])
            else:
                return None
    if shape is None or item is None or raw is None:
        return None
    count = 1
    for d in shape:
        count *= d
    return shape, item, raw[: count * item]


def scan(path):
    """``{(leaf name, sha256 hex, shape tuple, item size)}`` of every fixed-size dataset a symbol-table node links to."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"\x89HDF\r\n\x1a\n" and buf[8] == 0 and buf[13] == 8 and buf[14] == 8, "superblock 0, 8-byte offsets"

    def positions(sig):
        pos = buf.find(sig)
        while pos >= 0:
            if pos % 8 == 0:
                yield pos
            pos = buf.find(sig, pos + 4)

    heap_data = {}                                             # heap address -> (data segment address, size)
    for pos in positions(b"HEAP"):
        if buf[pos + 4] == 0:
            size, _free, data = _u("QQQ", buf, pos + 8)
            if data + size <= len(buf):
                heap_data[pos] = (data, size)
    nodes = {}                                                 # SNOD address -> [(name offset, header address, cache type, scratch)]
    for pos in positions(b"SNOD"):
        if buf[pos + 4] == 1:
            n = _u("H", buf, pos + 6)[0]
            nodes[pos] = [(*_u("QQI", buf, pos + 8 + 40 * k), bytes(buf[pos + 32 + 40 * k: pos + 48 + 40 * k])) for k in range(n)]
    heap_of_tree = {}                                          # B-tree root address -> its group's heap address
    root = _u("QQI", buf, 56)
    if root[2] == 1:
        heap_of_tree[_u("Q", buf, 56 + 24)[0]] = _u("Q", buf, 56 + 32)[0]
    for entries in nodes.values():
        for _name, _header, cache, scratch in entries:
            if cache == 1:
                tree, heap = _u("QQ", scratch, 0)
                heap_of_tree[tree] = heap
            else:                                              # not cached in the link: the group's own header says it
                for mtype, body in _messages(buf, _header):
                    if mtype == 0x0011:                        # symbol table message: B-tree address, heap address
                        tree, heap = _u("QQ", body, 0)
                        heap_of_tree[tree] = heap
    children = {}                                              # TREE node address -> (level, child addresses)
    for pos in positions(b"TREE"):
        if buf[pos + 4] == 0:                                  # node type 0: group node
            level, used = buf[pos + 5], _u("H", buf, pos + 6)[0]
            children[pos] = (level, [_u("Q", buf, pos + 24 + 8 + 16 * k)[0] for k in range(used)])
    heap_of_node = {}
    pending = list(heap_of_tree.items())
    while pending:
        tree, heap = pending.pop()
        level, kids = children.get(tree, (0, []))
        for kid in kids:
            if level == 0:
                heap_of_node[kid] = heap
            else:
                pending.append((kid, heap))

    found = set()
    for pos, entries in nodes.items():
        if pos not in heap_of_node:                            # a node no B-tree lists: freed space the library left behind
            continue
        data, size = heap_data[heap_of_node[pos]]
        for name_off, header, _cache, _scratch in entries:
            end = buf.find(b"\0", data + name_off, data + size)
            name = buf[data + name_off: end].decode()
            ds = _dataset(buf, header)
            if ds is not None:
                shape, item, raw = ds
                found.add((name, hashlib.sha256(raw).hexdigest(), shape, item))
    return found
