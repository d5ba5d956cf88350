"""TensorBoard event files of a fit: ``<output_path>/logs/<model>_run_<n>`` (the reference's
``SummaryWriter(log_dir=...)``, /root/reference/deepof/clustering/training.py:977-982, filled per epoch by
``log_epoch_to_tensorboard``, logging.py:436-467: ``Train/<term>``, ``Val/<term>``, ``Distill/lambda``,
``Val/alignment_score`` / ``conf_norm`` / ``bal_norm`` when the score is finite, ``Pretrain/total_loss``).

The ``tensorboard`` package is an optional dependency of the reference and is absent from the ROCm image, so the event
file is written directly: TFRecord framing (length, masked CRC-32C, payload, masked CRC-32C) around hand-encoded
``Event`` protocol buffers (wall_time = 1: double, step = 2: int64, file_version = 3: string, summary = 5:
{value = 1: {tag = 1: string, simple_value = 2: float}}).  TensorBoard reads these files as it reads SummaryWriter's.
"""
from __future__ import annotations

import math
import os
import socket
import struct
import time
from typing import Dict, Optional

_CRC_TABLE = []


def _crc32c(data: bytes) -> int:
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time: float, step: int, *, file_version: Optional[str] = None, tag: Optional[str] = None,
           value: float = 0.0) -> bytes:
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(step)
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if tag is not None:
        val = _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", value)
        ev += _field_bytes(5, _field_bytes(1, val))
    return ev


class EventFileWriter:
    """add_scalar / flush / close of torch.utils.tensorboard.SummaryWriter, scalars only."""

    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self.path = os.path.join(log_dir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}.0")
        self._f = open(self.path, "wb")
        self._write(_event(time.time(), 0, file_version="brain.Event:2"))

    def _write(self, payload: bytes):
        head = struct.pack("<Q", len(payload))
        self._f.write(head + struct.pack("<I", _masked_crc(head)) + payload + struct.pack("<I", _masked_crc(payload)))

    def add_scalar(self, tag: str, value, step: int):
        self._write(_event(time.time(), int(step), tag=tag, value=float(value)))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.flush()
            self._f.close()


def read_scalars(path: str):
    """[(step, tag, value)] of an event file written above (tests; checks framing and both checksums)."""
    out = []
    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        (n,) = struct.unpack_from("<Q", data, pos)
        assert struct.unpack_from("<I", data, pos + 8)[0] == _masked_crc(data[pos:pos + 8])
        payload = data[pos + 12:pos + 12 + n]
        assert struct.unpack_from("<I", data, pos + 12 + n)[0] == _masked_crc(payload)
        pos += 16 + n
        step, tag, value, p = 0, None, None, 0

        def varint(buf, q):
            v, shift = 0, 0
            while True:
                b = buf[q]
                q += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    return v, q

        while p < len(payload):
            key, p = varint(payload, p)
            num, wt = key >> 3, key & 7
            if wt == 1:
                p += 8
            elif wt == 0:
                v, p = varint(payload, p)
                if num == 2:
                    step = v
            elif wt == 2:
                ln, p = varint(payload, p)
                body = payload[p:p + ln]
                p += ln
                if num == 5:  # summary -> value -> (tag, simple_value)
                    _k, q = varint(body, 0)
                    ln2, q = varint(body, q)
                    val = body[q:q + ln2]
                    r = 0
                    while r < len(val):
                        k2, r = varint(val, r)
                        if k2 & 7 == 2:
                            l3, r = varint(val, r)
                            tag = val[r:r + l3].decode()
                            r += l3
                        elif k2 & 7 == 5:
                            (value,) = struct.unpack_from("<f", val, r)
                            r += 4
        if tag is not None:
            out.append((step, tag, value))
    return out


def open_writer(common_cfg, model_name: str, is_main: bool):
    """The reference's writer policy (training.py:977-982): rank 0 only, only with ``log_history``."""
    if not (getattr(common_cfg, "log_history", False) and is_main):
        return None
    log_dir = os.path.join(common_cfg.output_path, "logs", f"{model_name}_run_{common_cfg.run}")
    w = EventFileWriter(log_dir)
    print(f"TensorBoard logs -> {log_dir}")
    return w


def log_epoch_to_tensorboard(writer, train_logs: Dict[str, float], val_logs: Dict[str, float], epoch: int,
                             score_value: float = float("nan"), lambda_d: float = 0.0):
    """logging.py:436-467."""
    if writer is None:
        return
    for k, v in train_logs.items():
        writer.add_scalar(f"Train/{k}", v, epoch)
    for k, v in val_logs.items():
        writer.add_scalar(f"Val/{k}", v, epoch)
    writer.add_scalar("Distill/lambda", lambda_d, epoch)
    if math.isfinite(score_value):
        writer.add_scalar("Val/alignment_score", score_value, epoch)
        writer.add_scalar("Val/conf_norm", val_logs.get("conf_norm", float("nan")), epoch)
        writer.add_scalar("Val/bal_norm", val_logs.get("bal_norm", float("nan")), epoch)
    writer.flush()
