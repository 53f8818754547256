// TEST INFRASTRUCTURE — CPU oracle. City-based topology and latency of the reference, restated:
//   tools.CSVLatencyReader            T/CSVLatencyReader.java:258-344   (the latency matrix from the ping measurements)
//   core.geoinfo.Geo / GeoAllCities / GeoAWS / CityInfo   C/geoinfo/*.java
//   core.NodeBuilder.NodeBuilderWithCity                  C/NodeBuilder.java:98-147
//   core.NetworkLatency.AwsRegionNetworkLatency / NetworkLatencyByCity / NetworkLatencyByCityWJitter
//                                                          C/NetworkLatency.java:86-233
// The measurements themselves are DATA (tests/golden/city_data.json, made by tests/golden/make_city_data.py from the
// reference's resource CSVs) handed in through orc_city_data_*; everything computed from them is restated here,
// including the java.util.HashMap iteration orders the reference's results hang on: cumulative city probabilities are
// accumulated in HashMap<String,...>.entrySet() order (C/geoinfo/Geo.java:12-20) and a node's city is the first
// entry, in the same kind of order, whose cumulative probability reaches the draw (C/NodeBuilder.java:128-139).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

// java.lang.String.hashCode over UTF-16 code units (the data is ASCII) and java.util.HashMap's spreading of it
inline int32_t jstring_hash(const std::string& s) {
  uint32_t h = 0;
  for (unsigned char c : s) h = 31u * h + c;
  return (int32_t)h;
}

// java.util.HashMap<String, V> as far as ITERATION ORDER goes (OpenJDK 8+ java/util/HashMap.java): table of buckets
// indexed by (h ^ h >>> 16) & (cap - 1), a new key appended to its bucket's chain, resize() doubling the table and
// splitting every chain into a low and a high chain that keep their relative order, iteration bucket by bucket.
// Chains of 8 and more become trees, whose iteration order this class does not model: it refuses them.
template <class V>
class JHashMap {
 public:
  struct Entry {
    std::string key;
    V value;
    uint32_t hash;
  };
  explicit JHashMap(int initialCapacity = 0) {  // new HashMap<>() / new HashMap<>(n): tableSizeFor(n)
    if (initialCapacity > 0) {
      int c = 1;
      while (c < initialCapacity) c <<= 1;
      threshold_ = c;  // (HashMap keeps the initial capacity in `threshold` until the first put)
    }
  }
  // new HashMap<>(Map m): putMapEntries sizes the table for m.size() / 0.75 + 1 and puts in m's iteration order
  static JHashMap copyOf(const JHashMap& m) {
    float ft = ((float)m.size() / 0.75f) + 1.0f;
    JHashMap r((int)ft);
    for (const Entry* e : m.entries()) r.put(e->key, e->value);
    return r;
  }
  void put(const std::string& key, const V& v) {
    const uint32_t h0 = (uint32_t)jstring_hash(key), h = h0 ^ (h0 >> 16);
    if (table_.empty()) resize();
    std::vector<int>& b = table_[h & (table_.size() - 1)];
    for (int i : b)
      if (store_[i].key == key) {
        store_[i].value = v;
        return;
      }
    store_.push_back({key, v, h});
    b.push_back((int)store_.size() - 1);
    if (b.size() >= 8) throw std::runtime_error("orc::JHashMap: a bucket of 8 entries would be treeified (not modelled)");
    if (++size_ > threshold_) resize();
  }
  const V* get(const std::string& key) const {
    if (table_.empty()) return nullptr;
    const uint32_t h0 = (uint32_t)jstring_hash(key), h = h0 ^ (h0 >> 16);
    for (int i : table_[h & (table_.size() - 1)])
      if (store_[i].key == key) return &store_[i].value;
    return nullptr;
  }
  bool containsKey(const std::string& key) const { return get(key) != nullptr; }
  void remove(const std::string& key) {
    if (table_.empty()) return;
    const uint32_t h0 = (uint32_t)jstring_hash(key), h = h0 ^ (h0 >> 16);
    std::vector<int>& b = table_[h & (table_.size() - 1)];
    for (size_t k = 0; k < b.size(); k++)
      if (store_[b[k]].key == key) {
        b.erase(b.begin() + (long)k);
        size_--;
        return;
      }
  }
  int size() const { return size_; }
  std::vector<const Entry*> entries() const {  // entrySet() iteration order
    std::vector<const Entry*> out;
    for (const auto& b : table_)
      for (int i : b) out.push_back(&store_[i]);
    return out;
  }

 private:
  void resize() {
    if (table_.empty()) {
      const int cap = threshold_ > 0 ? threshold_ : 16;
      table_.assign((size_t)cap, {});
      threshold_ = (int)(cap * 0.75f);
      return;
    }
    const size_t oldCap = table_.size();
    std::vector<std::vector<int>> nt(oldCap * 2);
    for (size_t j = 0; j < oldCap; j++)
      for (int i : table_[j]) nt[(store_[i].hash & oldCap) ? j + oldCap : j].push_back(i);
    table_.swap(nt);
    threshold_ *= 2;
  }
  std::vector<Entry> store_;
  std::vector<std::vector<int>> table_;
  int size_ = 0, threshold_ = 0;
};

struct CityInfo {  // C/geoinfo/CityInfo.java
  int mercX, mercY;
  float cumulativeProbability;
};

// Geo.cityInfoMap (C/geoinfo/Geo.java:11-21): cumulative probabilities in the iteration order of `cities`
inline JHashMap<CityInfo> cityInfoMap(const JHashMap<std::vector<int>>& cities, int totalPopulation) {
  float cumulativeProbability = 0.f;
  JHashMap<CityInfo> citiesInfo;
  for (const auto* e : cities.entries()) {
    cumulativeProbability = cumulativeProbability + (float)e->value[2] * 1.f / (float)totalPopulation;
    citiesInfo.put(e->key, CityInfo{e->value[0], e->value[1], cumulativeProbability});
  }
  return citiesInfo;
}

inline long jround(double x) { return (long)std::floor(x + 0.5); }  // Math.round(double)
inline int jround_f(float x) { return (int)std::floor((double)x + 0.5); }  // Math.round(float): floor(x + 1/2), exact

// The data set (orc_city_data_*): CSVLatencyReader's city list, per city the parsed measurements, cities.csv rows.
struct CityData {
  std::vector<std::string> dirs;                         // T/CSVLatencyReader.java:14-257
  std::vector<std::map<std::string, std::string>> ping;  // latenciesForCity, values as the CSV's strings
  std::vector<std::vector<std::string>> cityRows;        // cities.csv: name, Lat, Long, Population
  bool loaded() const { return !dirs.empty(); }
};
inline CityData& cityData() {
  static CityData d;
  return d;
}

// tools.CSVLatencyReader: makeLatencyMatrix (:303-312), then the cities with a missing pair removed (:285-290, :336-350)
struct CSVLatencyReader {
  JHashMap<JHashMap<float>> latencyMatrix;
  CSVLatencyReader() {
    const CityData& d = cityData();
    if (!d.loaded()) throw std::runtime_error("city data not loaded (orc_city_data_*)");
    for (size_t i = 0; i < d.dirs.size(); i++) {
      JHashMap<float> m;
      // (a java.util.HashMap filled in CSV row order; only lookups are made on it, so the order does not matter)
      for (const auto& kv : d.ping[i]) m.put(kv.first, strtof(kv.second.c_str(), nullptr));  // Float.valueOf
      m.put(d.dirs[i], 30.f);  // SAME_CITY_LATENCY :12
      latencyMatrix.put(d.dirs[i], m);
    }
    std::vector<std::string> missing;
    const auto all = latencyMatrix.entries();
    for (const auto* from : all)
      for (const auto* to : all)
        if (!from->value.containsKey(to->key) && !to->value.containsKey(from->key)) {
          missing.push_back(from->key);
          break;
        }
    for (const auto& c : missing) latencyMatrix.remove(c);
  }
  std::vector<std::string> cities() const {  // new ArrayList<>(latencyMatrix.keySet())
    std::vector<std::string> out;
    for (const auto* e : latencyMatrix.entries()) out.push_back(e->key);
    return out;
  }
};

inline std::string upper(const std::string& s) {
  std::string r = s;
  for (auto& c : r) c = (char)toupper((unsigned char)c);
  return r;
}

// C/geoinfo/GeoAllCities.java:31-80
inline JHashMap<CityInfo> geoAllCitiesPosition() {
  const CityData& d = cityData();
  const double mapWidth = 2000, mapHeight = 1112;  // Node.MAX_X / MAX_Y
  JHashMap<std::vector<int>> cities;
  int totalPopulation = 0;
  for (const auto& r : d.cityRows) {
    std::string cityName = r[0];
    for (auto& c : cityName)
      if (c == ' ') c = '+';
    const float latitude = strtof(r[1].c_str(), nullptr), longitude = strtof(r[2].c_str(), nullptr);
    int posX = (int)(((double)longitude + 180) * (mapWidth / 360));  // convertToMercatorX(double)
    posX = posX < mapWidth / 2 ? posX - 45 : posX - 70;
    int posY = (int)jround((mapHeight / 2) - ((double)latitude * mapHeight / 180));  // convertToMercatorY(float)
    if (posY < 0.2 * mapHeight) posY = posY - 35;
    int population = atoi(r[3].c_str());
    population += 200000;
    totalPopulation += population;
    cities.put(cityName, {posX, posY, population});
  }
  const JHashMap<CityInfo> citiesPosition = cityInfoMap(cities, totalPopulation);  // the field, built once (:28)
  return JHashMap<CityInfo>::copyOf(citiesPosition);                                // citiesPosition() :31-33
}

// C/geoinfo/GeoAWS.java:9-28
inline JHashMap<CityInfo> geoAwsPosition() {
  JHashMap<std::vector<int>> cities;
  cities.put("Oregon", {271, 261, 1});
  cities.put("Virginia", {513, 316, 1});
  cities.put("Mumbai", {1344, 426, 1});
  cities.put("Seoul", {1641, 312, 1});
  cities.put("Singapore", {1507, 532, 1});
  cities.put("Sydney", {1773, 777, 1});
  cities.put("Tokyo", {1708, 316, 1});
  cities.put("Canada central", {422, 256, 1});
  cities.put("Frankfurt", {985, 226, 1});
  cities.put("Ireland", {891, 200, 1});
  cities.put("London", {937, 205, 1});
  return cityInfoMap(cities, cities.size());
}

inline const std::vector<std::string>& awsRegions() {  // regionPerCity order = region index (C/NetworkLatency.java:90-102)
  static const std::vector<std::string> r = {"Oregon", "Virginia", "Mumbai", "Seoul", "Singapore", "Sydney",
                                             "Tokyo", "Canada central", "Frankfurt", "Ireland", "London"};
  return r;
}

// NodeBuilder.NodeBuilderWithCity (C/NodeBuilder.java:98-147) as a table: citiesInfo in its entrySet() order
struct CityChooser {
  std::vector<std::string> name;
  std::vector<CityInfo> info;
  int listSize = 0;  // cities.size(): the LIST the builder was given, not the filtered map
  CityChooser(const std::vector<std::string>& cities, const JHashMap<CityInfo>& geo) {
    std::vector<std::string> up;
    for (const auto& c : cities) up.push_back(upper(c));
    listSize = (int)up.size();
    JHashMap<CityInfo> citiesInfo;  // Collectors.toMap: a new HashMap filled in the stream's (= geo's iteration) order
    for (const auto* e : geo.entries()) {
      bool in = false;
      for (const auto& c : up) in |= c == upper(e->key);
      if (in) citiesInfo.put(e->key, e->value);
    }
    for (const auto* e : citiesInfo.entries()) {
      name.push_back(e->key);
      info.push_back(e->value);
    }
  }
  int choose(int32_t rdInt) const {  // getRandomCityInfo :128-139; -1 = null
    const int size = listSize;
    const int rand = (rdInt == INT32_MIN ? INT32_MIN : std::abs(rdInt)) % size;  // Math.abs(int), Java %
    const float p = (float)rand / (float)size;
    for (size_t i = 0; i < info.size(); i++)
      if (p <= info[i].cumulativeProbability) return (int)i;
    return -1;
  }
};

}  // namespace orc
