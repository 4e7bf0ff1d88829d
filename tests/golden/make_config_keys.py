"""Writes tests/golden/reference_yaml_keys.json: for every yaml under /root/reference/configs the flattened {key: value} map of what it SETS
(own keys only, `_BASE_` recorded separately).  Data of the reference's config files, not source; regenerate with
`python tests/golden/make_config_keys.py` in a container that has /root/reference."""
import json
import os

import yaml

REF = "/root/reference/configs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_yaml_keys.json")


def flat(d, pre=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flat(v, pre + k + "."))
        else:
            out[pre + k] = list(v) if isinstance(v, tuple) else v
    return out


def main():
    res = {}
    for root, _, files in os.walk(REF):
        for f in sorted(files):
            if f.endswith(".yaml"):
                p = os.path.join(root, f)
                d = yaml.unsafe_load(open(p)) or {}
                base = d.pop("_BASE_", None)
                res[os.path.relpath(p, REF)] = {"_BASE_": base, "keys": flat(d)}
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print(OUT, len(res), "yaml files")


if __name__ == "__main__":
    main()
