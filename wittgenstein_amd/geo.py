"""City topologies and city latency models for the engine: the host-side computation the reference performs in
RegistryNodeBuilders / NodeBuilderWithCity / GeoAllCities / GeoAWS / CSVLatencyReader and the three city-based
NetworkLatency classes (C/RegistryNodeBuilders.java:44-58, C/NodeBuilder.java:98-147, C/geoinfo/*.java,
T/CSVLatencyReader.java:258-344, C/NetworkLatency.java:86-233), producing the plain tables the C ABI takes
(wgh_register_city_builder / wgh_register_city_latency, wg_set_latency_city). A Java host computes the same tables
with the reference's own classes; this module is the stand-in where no JVM exists.

The measurements are the caller's data: `load(path)` reads the JSON written by tests/golden/make_city_data.py (city
list, parsed ping averages, cities.csv rows). What the reference's results hang on beyond the numbers is restated
here: float32 arithmetic of the cumulative city probabilities, and the iteration order of java.util.HashMap<String,
...> in which they are accumulated and searched (`JavaHashMap`)."""
import ctypes as C
import json
import math

import numpy as np

from . import _lib as L

F32 = np.float32


def _jhash(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h ^ (h >> 16)


class JavaHashMap:
    """iteration order of java.util.HashMap<String, V> (OpenJDK 8+): buckets by (h ^ h >>> 16) & (cap - 1), chains in
    insertion order, resize() splitting a chain into a low and a high half that keep their order. Chains that would be
    treeified (8 entries) are refused."""

    def __init__(self, initial_capacity=0):
        self.table, self.size, self.threshold, self.vals = None, 0, 0, {}
        if initial_capacity > 0:
            c = 1
            while c < initial_capacity:
                c <<= 1
            self.threshold = c

    @classmethod
    def copy_of(cls, m):  # new HashMap<>(m): putMapEntries
        r = cls(int(F32(F32(m.size) / F32(0.75)) + F32(1.0)))
        for k, v in m.items():
            r.put(k, v)
        return r

    def _resize(self):
        if self.table is None:
            cap = self.threshold if self.threshold > 0 else 16
            self.table = [[] for _ in range(cap)]
            self.threshold = int(cap * 0.75)
            return
        old = len(self.table)
        nt = [[] for _ in range(2 * old)]
        for j, b in enumerate(self.table):
            for k in b:
                nt[j + old if (_jhash(k) & old) else j].append(k)
        self.table = nt
        self.threshold *= 2

    def put(self, k, v):
        if self.table is None:
            self._resize()
        if k in self.vals:
            self.vals[k] = v
            return
        b = self.table[_jhash(k) & (len(self.table) - 1)]
        b.append(k)
        if len(b) >= 8:
            raise RuntimeError("JavaHashMap: a bucket of 8 entries would be treeified (not modelled)")
        self.vals[k] = v
        self.size += 1
        if self.size > self.threshold:
            self._resize()

    def remove(self, k):
        if k in self.vals:
            self.table[_jhash(k) & (len(self.table) - 1)].remove(k)
            del self.vals[k]
            self.size -= 1

    def get(self, k):
        return self.vals.get(k)

    def __contains__(self, k):
        return k in self.vals

    def keys(self):
        return [k for b in (self.table or []) for k in b]

    def items(self):
        return [(k, self.vals[k]) for k in self.keys()]


def load(path):
    return json.load(open(path))


def latency_matrix(data):
    """CSVLatencyReader(): makeLatencyMatrix, then the cities with a missing pair removed (T/CSVLatencyReader.java:
    285-290,303-312,336-350). Returns the JavaHashMap city -> {other city -> float32}."""
    m = JavaHashMap()
    for city, row in zip(data["dirs"], data["ping"]):
        d = {k: F32(v) for k, v in row.items()}  # Float.valueOf
        d[city] = F32(30.0)  # SAME_CITY_LATENCY
        m.put(city, d)
    keys = m.keys()
    missing = [a for a in keys if any(b not in m.get(a) and a not in m.get(b) for b in keys)]
    for a in missing:
        m.remove(a)
    return m


def _city_info_map(cities, total_population):  # Geo.cityInfoMap (C/geoinfo/Geo.java:11-21)
    cum = F32(0.0)
    out = JavaHashMap()
    for k, (x, y, pop) in cities.items():
        cum = F32(cum + F32(F32(F32(pop) * F32(1.0)) / F32(total_population)))
        out.put(k, (x, y, cum))
    return out


def _jround(x):  # Math.round(double)
    return math.floor(x + 0.5)


def geo_all_cities(data):
    """new GeoAllCities().citiesPosition() (C/geoinfo/GeoAllCities.java:22-80)"""
    w, h = 2000.0, 1112.0
    cities = JavaHashMap()
    total = 0
    for name, lat, lon, pop in data["cities"]:
        lat, lon = F32(lat), F32(lon)
        px = int((float(lon) + 180) * (w / 360))
        px = px - 45 if px < w / 2 else px - 70
        py = int(_jround((h / 2) - (float(lat) * h / 180)))
        if py < 0.2 * h:
            py -= 35
        p = int(pop) + 200000
        total += p
        cities.put(name.replace(" ", "+"), (px, py, p))
    return JavaHashMap.copy_of(_city_info_map(cities, total))


AWS_REGIONS = ["Oregon", "Virginia", "Mumbai", "Seoul", "Singapore", "Sydney", "Tokyo", "Canada central", "Frankfurt",
               "Ireland", "London"]  # regionPerCity's indices (C/NetworkLatency.java:90-102)
AWS_POS = [(271, 261), (513, 316), (1344, 426), (1641, 312), (1507, 532), (1773, 777), (1708, 316), (422, 256),
           (985, 226), (891, 200), (937, 205)]  # C/geoinfo/GeoAWS.java:12-22
AWS_PING = [[0, 81, 216, 126, 165, 138, 97, 64, 164, 131, 141], [0, 0, 182, 181, 232, 195, 167, 13, 88, 80, 75],
            [0, 0, 0, 152, 62, 223, 123, 194, 111, 122, 113], [0, 0, 0, 0, 97, 133, 35, 184, 259, 254, 264],
            [0, 0, 0, 0, 0, 169, 69, 218, 162, 174, 171], [0, 0, 0, 0, 0, 0, 105, 210, 282, 269, 271],
            [0, 0, 0, 0, 0, 0, 0, 156, 235, 222, 234], [0, 0, 0, 0, 0, 0, 0, 0, 101, 78, 87],
            [0, 0, 0, 0, 0, 0, 0, 0, 0, 24, 13], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12]]  # C/NetworkLatency.java:113-133


def geo_aws():
    cities = JavaHashMap()
    for name, (x, y) in zip(AWS_REGIONS, AWS_POS):
        cities.put(name, (x, y, 1))
    return _city_info_map(cities, cities.size)


class NodeBuilderWithCity:
    """C/NodeBuilder.java:98-147 as a table: citiesInfo (a Collectors.toMap HashMap filled in the geo map's iteration
    order, filtered by the upper-cased city list) in its own entrySet() order."""

    def __init__(self, cities, geo):
        up = {c.upper() for c in cities}
        self.list_size = len(cities)
        info = JavaHashMap()
        for k, v in geo.items():
            if k.upper() in up:
                info.put(k, v)
        it = info.items()
        self.names = [k for k, _ in it]
        self.merc_x = np.array([v[0] for _, v in it], np.int32)
        self.merc_y = np.array([v[1] for _, v in it], np.int32)
        self.cum = np.array([v[2] for _, v in it], np.float32)

    def choose(self, rd_int):
        """getRandomCityInfo (:128-139): the row of the node's city, None where the reference returns null"""
        a = rd_int if rd_int == -2**31 else abs(rd_int)
        rand = int(math.fmod(a, self.list_size))  # Java % truncates
        p = F32(F32(rand) / F32(self.list_size))
        hit = np.nonzero(p <= self.cum)[0]
        return int(hit[0]) if len(hit) else None


def gpd_jitter():
    """gpd.inverseF(delta / 100.0), delta 0..99, of GeneralizedParetoDistribution(1.4, -0.3, 0.35)
    (C/utils/GeneralizedParetoDistribution.java:26-46; C/NetworkLatency.java:53,205)"""
    shape, location, scale = 1.4, -0.3, 0.35
    out = np.zeros(100, np.float64)
    for d in range(100):
        y = d / 100.0
        out[d] = location if y < 0.000001 else location + scale / shape * (-1 + math.pow(1 - y, -shape))
    return out


def aws_tables(builder):
    """wg_set_latency_city(WG_CITY_AWS) tables over the builder's rows"""
    n = len(builder.names)
    reg = [AWS_REGIONS.index(c) for c in builder.names]
    tab = np.zeros((n, n), np.int32)
    for i in range(n):
        for j in range(n):
            if reg[i] != reg[j]:
                tab[i, j] = AWS_PING[min(reg[i], reg[j])][max(reg[i], reg[j])] // 2
    return tab, None, gpd_jitter()


def city_tables(builder, matrix):
    """(tab for WG_CITY_BY_CITY, ping for WG_CITY_BY_CITY_WJITTER, jitter) over the builder's rows;
    getLatency(cityFrom, cityTo) with its to->from fallback (C/NetworkLatency.java:187-197)"""
    n = len(builder.names)
    ping = np.zeros((n, n), np.float32)
    for i, a in enumerate(builder.names):
        ra = matrix.get(a)
        for j, b in enumerate(builder.names):
            v = ra.get(b)
            ping[i, j] = v if v is not None else matrix.get(b)[a]
    half = (F32(0.5) * ping).astype(np.float32)
    tab = np.maximum(1, np.floor(half.astype(np.float64) + 0.5)).astype(np.int32)  # max(1, Math.round(0.5f * ping))
    return tab, ping, gpd_jitter()


_REGISTERED = {}


def register(path):
    """compute the AWS and CITIES builder tables and the three city latency models from the data file and hand them to
    the library: afterwards the protocol mirrors accept nodeBuilderName "AWS_..." / "CITIES_..." and networkLatencyName
    "AwsRegionNetworkLatency" / "NetworkLatencyByCity" / "NetworkLatencyByCityWJitter". Returns the two builders."""
    if path in _REGISTERED:
        return _REGISTERED[path]
    data = load(path)
    matrix = latency_matrix(data)
    aws = NodeBuilderWithCity(sorted(AWS_REGIONS), geo_aws())
    cities = NodeBuilderWithCity(matrix.keys(), geo_all_cities(data))
    lib = L.lib()

    def p(a, t):
        return None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.POINTER(t))

    def ck(rc):
        if rc != L.WG_OK:
            raise RuntimeError(lib.wgh_last_error().decode())
    for site, b in (("AWS", aws), ("CITIES", cities)):
        ck(lib.wgh_register_city_builder(site.encode(), len(b.names), p(b.cum, C.c_float), p(b.merc_x, C.c_int32),
                                         p(b.merc_y, C.c_int32), b.list_size))
    at, _, aj = aws_tables(aws)
    ck(lib.wgh_register_city_latency(b"AwsRegionNetworkLatency", 0, len(aws.names), p(at, C.c_int32), None, p(aj, C.c_double)))
    ct, cp, cj = city_tables(cities, matrix)
    ck(lib.wgh_register_city_latency(b"NetworkLatencyByCity", 1, len(cities.names), p(ct, C.c_int32), None, None))
    ck(lib.wgh_register_city_latency(b"NetworkLatencyByCityWJitter", 2, len(cities.names), None, p(cp, C.c_float), p(cj, C.c_double)))
    _REGISTERED[path] = {"AWS": aws, "CITIES": cities, "aws_tables": (at, None, aj), "city_tables": (ct, cp, cj)}
    return _REGISTERED[path]
