import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    sp=d.get("sorted_pipeline") or {}
    print(d.get("config","")[:28], "wall", round(d.get("wall_us",0),1), "sort", sp.get("us_sort"), "walk", sp.get("us_walk"))
