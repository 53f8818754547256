#!/usr/bin/env python
"""Generates tests/golden/city_data.json — the DATA the reference's city latency models and city node builders read —
from the reference's resource files (core/src/main/resources/cities.csv and Data/<City>/<City>Ping.csv, the
wondernetwork.com ping measurements). Data only: no reference source text is copied. What the fixture holds:

  dirs      the city list of T/CSVLatencyReader.java:14-257 (= the Data/ directory names, in that list's order)
  ping      per city (same order) the map {other city -> Average ms string} that CSVLatencyReader.latenciesForCity
            (:318-334) builds: column 4 of the CSV ("hidden.1") keyed by processCityName (:346-356: the LONGEST city of
            the list whose name, '+' as ' ', is contained in the row's "City" cell), later rows overwriting earlier ones
  cities    cities.csv rows as strings: name, Lat, Long, Population (C/geoinfo/GeoAllCities.java:31-56 parses them)

Everything downstream of that — SAME_CITY_LATENCY, the removal of cities with missing measurements, the to->from
fallback, Float parsing, the Mercator integers, HashMap iteration orders, the latency formulas — is restated in
oracle/geo.hpp and in wittgenstein_amd/geo.py and is NOT precomputed here.
Run in the build container (needs /root/reference):  python tests/golden/make_city_data.py"""
import csv
import json
import os
import re

REF = "/root/reference/core/src/main"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(os.path.join(REF, "java/net/consensys/wittgenstein/tools/CSVLatencyReader.java")).read()
    block = src[src.index("Arrays.asList("):src.index(");", src.index("Arrays.asList("))]
    dirs = re.findall(r'"([^"]+)"', block)
    assert sorted(os.listdir(os.path.join(REF, "resources/Data"))) == sorted(dirs), "Data/ directories != the list"
    plain = [(c, c.replace("+", " ")) for c in dirs]
    ping = []
    for city in dirs:
        m = {}
        with open(os.path.join(REF, "resources/Data", city, city + "Ping.csv"), newline="") as f:
            rows = csv.reader(f)
            next(rows)  # CSVFormat.DEFAULT.withHeader()
            for r in rows:
                hits = [c for c, p in plain if p in r[0]]
                if hits:
                    best = hits[0]
                    for c in hits[1:]:  # Stream.max = reduce(BinaryOperator.maxBy): (a, b) -> cmp(a, b) >= 0 ? a : b,
                        if len(c) > len(best):  # i.e. the FIRST of equal maxima stays
                            best = c
                    m[best] = r[4]
        ping.append(m)
    with open(os.path.join(REF, "resources/cities.csv"), newline="") as f:
        rows = list(csv.reader(f))[1:]
    out = {"dirs": dirs, "ping": ping, "cities": [r[:4] for r in rows]}
    with open(os.path.join(HERE, "city_data.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", len(dirs), "cities,", sum(len(m) for m in ping), "measurements,", len(rows), "cities.csv rows")


if __name__ == "__main__":
    main()
