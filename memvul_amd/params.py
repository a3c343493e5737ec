"""Config loading for the hot path: the reference's JSON overrides (test_config_memory.json) and its
Jsonnet training configs (MemVul/config_memory.json: ``local`` string/number variables, trailing commas),
merged the way ``load_archive(..., overrides=...)`` does (predict_memory.py:60-67): a recursive dict
update of the archived config by the override dict."""
from __future__ import annotations

import copy
import json
import re
from typing import Any, Dict, Union

_LOCAL_RE = re.compile(r"^\s*local\s+([A-Za-z_]\w*)\s*=\s*(.+?);\s*$", re.M)


def parse_jsonnet_subset(text: str) -> Dict[str, Any]:
    """JSON, or the Jsonnet subset the reference configs use: top-level ``local name = <json scalar>;``
    bindings referenced as bare identifiers in value position, ``//`` comments, trailing commas."""
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        pass
    text = re.sub(r"(^|\s)//[^\n]*", r"\1", text)
    binds = {m.group(1): m.group(2).strip() for m in _LOCAL_RE.finditer(text)}
    text = _LOCAL_RE.sub("", text)
    out, i, n = [], 0, len(text)
    while i < n:  # substitute identifiers outside string literals
        c = text[i]
        if c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append(text[i : j + 1])
            i = j + 1
        elif c.isalpha() or c == "_":
            j = i
            while j < n and (text[j].isalnum() or text[j] == "_"):
                j += 1
            word = text[i:j]
            out.append(binds.get(word, word))
            i = j
        else:
            out.append(c)
            i += 1
    text = "".join(out)
    text = re.sub(r",(\s*[}\]])", r"\1", text)
    return json.loads(text)


def load_config(path_or_dict: Union[str, Dict[str, Any]]) -> Dict[str, Any]:
    if isinstance(path_or_dict, dict):
        return copy.deepcopy(path_or_dict)
    with open(path_or_dict, "r", encoding="utf-8") as f:
        return parse_jsonnet_subset(f.read())


def with_overrides(config: Dict[str, Any], overrides: Union[str, Dict[str, Any], None]) -> Dict[str, Any]:
    """AllenNLP ``with_fallback``: override keys win, dicts merge recursively."""
    if not overrides:
        return copy.deepcopy(config)
    if isinstance(overrides, str):
        overrides = json.loads(overrides)

    def merge(base, over):
        out = copy.deepcopy(base)
        for k, v in over.items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = merge(out[k], v)
            else:
                out[k] = copy.deepcopy(v)
        return out

    return merge(config, overrides)
