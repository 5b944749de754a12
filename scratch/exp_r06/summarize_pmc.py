#!/usr/bin/env python3
"""rocprofv3 --pmc output directory -> one small JSON: for every dispatch of the rollout kernel its duration (the tool's own timestamps) and
its counters; with the json output format present, also the per-instance values of the counters that have instances (TCC channels)."""
import csv
import glob
import gzip
import json
import os
import sys

d, out = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "gpd_rollout1"
rows = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = rows.setdefault(k, {"dispatch": k, "grid": int(r["Grid_Size"]), "counters": {}})
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        e["counters"][r["Counter_Name"]] = e["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
res = {"kernel": want, "dispatches": [rows[k] for k in sorted(rows)]}
# per-instance values from the json output (when asked for): kept raw but small -- only this kernel's dispatches
for f in glob.glob(os.path.join(d, "**", "*results.json"), recursive=True):
    try:
        j = json.load(open(f))["rocprofiler-sdk-tool"][0]
        names = {}
        for c in j.get("counters", []):
            names[c["id"]["handle"] if isinstance(c.get("id"), dict) else c.get("id")] = c
        ksym = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in j.get("kernel_symbols", [])}
        per = []
        for rec in j.get("callback_records", {}).get("counter_collection", []):
            info = rec.get("dispatch_data", {}).get("dispatch_info", {})
            if want not in ksym.get(info.get("kernel_id"), ""):
                continue
            per.append({"dispatch": info.get("dispatch_id"), "records": rec.get("records", [])})
        res["json_counter_names"] = {str(k): {"name": v.get("name"), "dimension_ids": v.get("dimension_ids")} for k, v in names.items() if v.get("name", "").startswith(("TCC_", "TCP_"))}
        res["json_dimensions"] = j.get("dimensions") or [dict(id=x.get("id"), name=x.get("name"), instance_size=x.get("instance_size")) for c in j.get("counters", [])[:1] for x in c.get("dimensions", [])]
        res["per_instance"] = per
    except Exception as e:      # noqa: BLE001
        res["json_error"] = f"{type(e).__name__}: {e}"
    os.remove(f)                # (hundreds of MB with every torch kernel's records)
with gzip.open(out, "wt") as fh:
    json.dump(res, fh)
print(out, len(res["dispatches"]), "dispatches", os.path.getsize(out), "bytes")
