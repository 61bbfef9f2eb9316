"""nominal.ini surface: an inih-compatible reader and the parameter sets of the three stages.

Behaviour follows the reference's vendored inih + INIReader, including the quirks that change results
(SURVEY.md 5.6):
  * an inline ';' starts a comment only when preceded by whitespace      (src/lib/ini.c:43-52)
    so `use_qv = true;` has the value "true;" and GetBoolean falls back to its default
  * integers parse by strtol prefix ("1000;" -> 1000)                    (src/lib/INIReader.cpp:32-40)
  * lines are read with fgets into 200 bytes, continuation lines (leading whitespace) append with '\n'
  * filter reads `[layout] del_telomere`, layout reads `[layout] del_telomeres`
                                                   (src/filter/filter.cpp:406, src/layout/hinging.cpp:803)
  * `hinge_bin` is overwritten by 2 * hinge_tolerance_length             (src/filter/filter.cpp:401-405)
"""
from __future__ import annotations

import re
from typing import Dict

from .capi import FilterParams

_WS = " \t\n\r\x0b\x0c"


def _find_char_or_comment(s: str, start: int, c: str) -> int:
    was_ws = False
    i = start
    while i < len(s) and s[i] != c and not (was_ws and s[i] == ";"):
        was_ws = s[i] in _WS
        i += 1
    return i


class IniFile:
    def __init__(self, path: str):
        self.values: Dict[str, str] = {}
        self.error = 0
        try:
            with open(path, "rb") as f:
                data = f.read()
        except OSError:
            self.error = -1
            return
        # fgets(line, 200): at most 199 bytes per "line"
        lines = []
        for raw in data.split(b"\n"):
            raw = raw + b"\n"
            while len(raw) > 199:
                lines.append(raw[:199])
                raw = raw[199:]
            lines.append(raw)
        if lines and lines[-1] == b"\n":
            lines.pop()
        section, prev_name = "", ""
        for lineno, rawb in enumerate(lines, 1):
            line = rawb.decode("latin-1")
            if lineno == 1 and line.startswith("\xef\xbb\xbf"):
                line = line[3:]
            stripped = line.rstrip(_WS)
            start = len(stripped) - len(stripped.lstrip(_WS))
            body = stripped[start:]
            if body[:1] in (";", "#"):
                continue
            if prev_name and body and start > 0:
                self._handle(section, prev_name, body)
            elif body[:1] == "[":
                end = _find_char_or_comment(body, 1, "]")
                if end < len(body) and body[end] == "]":
                    section = body[1:end][:49]
                    prev_name = ""
                elif not self.error:
                    self.error = lineno
            elif body:
                end = _find_char_or_comment(body, 0, "=")
                if not (end < len(body) and body[end] == "="):
                    end = _find_char_or_comment(body, 0, ":")
                if end < len(body) and body[end] in "=:":
                    name = body[:end].rstrip(_WS)
                    value = body[end + 1:].lstrip(_WS)
                    cend = _find_char_or_comment(value, 0, "\0")
                    if cend < len(value) and value[cend] == ";":
                        value = value[:cend]
                    value = value.rstrip(_WS)
                    prev_name = name[:49]
                    self._handle(section, name, value)
                elif not self.error:
                    self.error = lineno

    def _handle(self, section: str, name: str, value: str):
        k = (section + "=" + name).lower()
        if self.values.get(k):
            self.values[k] += "\n"
        self.values[k] = self.values.get(k, "") + value

    def get(self, section: str, name: str, default: str) -> str:
        return self.values.get((section + "=" + name).lower(), default)

    def get_int(self, section: str, name: str, default: int) -> int:
        v = self.get(section, name, "")
        m = re.match(r"[ \t\n\r\x0b\x0c]*([+-]?)(0[xX][0-9a-fA-F]+|0[0-7]*|[1-9][0-9]*)", v)
        if not m:
            return default
        sign = -1 if m.group(1) == "-" else 1
        tok = m.group(2)
        if tok[:2].lower() == "0x":
            val = int(tok, 16)
        elif tok.startswith("0") and len(tok) > 1:
            val = int(tok, 8)
        else:
            val = int(tok, 10)
        return sign * val

    def get_real(self, section: str, name: str, default: float) -> float:
        v = self.get(section, name, "")
        m = re.match(r"[ \t\n\r\x0b\x0c]*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", v)
        return float(m.group(0)) if m else default

    def get_bool(self, section: str, name: str, default: bool) -> bool:
        v = self.get(section, name, "").lower()
        if v in ("true", "yes", "on", "1"):
            return True
        if v in ("false", "no", "off", "0"):
            return False
        return default


def filter_params(ini: IniFile, has_qv: bool) -> FilterParams:
    """filter.cpp:377-409."""
    p = FilterParams()
    p.reso = 40
    p.cut_off = ini.get_int("filter", "cut_off", -1)
    p.min_cov = ini.get_int("filter", "min_cov", -1)
    p.est_cov = ini.get_int("filter", "ec", 0)
    p.theta = ini.get_int("filter", "theta", -1)
    p.coverage_fraction = ini.get_int("filter", "coverage_frac_repeat_annotation", 3)
    p.min_repeat_annotation = ini.get_int("filter", "min_repeat_annotation_threshold", 10)
    p.max_repeat_annotation = ini.get_int("filter", "max_repeat_annotation_threshold", 20)
    p.repeat_annotation_gap = ini.get_int("filter", "repeat_annotation_gap_threshold", 300)
    p.no_hinge_region = ini.get_int("filter", "no_hinge_region", 500)
    p.hinge_min_support = ini.get_int("filter", "hinge_min_support", 7)
    p.hinge_bin_pileup = ini.get_int("filter", "hinge_min_pileup", 7)
    p.hinge_unbridged = ini.get_int("filter", "hinge_unbridged", 6)
    p.hinge_tolerance = ini.get_int("filter", "hinge_tolerance_length", 100)
    p.use_qv_mask = int(ini.get_bool("filter", "use_qv", True) and has_qv)
    p.use_coverage_mask = int(ini.get_bool("filter", "coverage", True))
    p.delete_telomere = int(bool(ini.get_int("layout", "del_telomere", 0)))
    return p


def default_filter_params() -> FilterParams:
    """utils/nominal.ini as the reference parses it (no qual track)."""
    p = FilterParams()
    (p.reso, p.cut_off, p.min_cov, p.est_cov, p.theta) = (40, 300, 5, 0, 300)
    (p.coverage_fraction, p.min_repeat_annotation, p.max_repeat_annotation, p.repeat_annotation_gap) = (3, 10, 20, 300)
    (p.no_hinge_region, p.hinge_min_support, p.hinge_bin_pileup, p.hinge_unbridged, p.hinge_tolerance) = (500, 7, 7, 6, 100)
    (p.use_qv_mask, p.use_coverage_mask, p.delete_telomere) = (0, 1, 0)
    return p
