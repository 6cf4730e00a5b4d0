// EnumWrapper.hpp -- configuring enum-valued options by name.
//
// Same surface as the reference's EnumWrap::EnumWrapper<E> (core/src/include/EnumWrapper.hpp:58-113): a per-enum static
// map from the strings of the configuration file to the enum values (setMap), operator()(key) to set the wrapped value,
// the conversion back to the plain enum, and a stream extraction operator -- which is all the option machinery needs:
// the reference hands the wrapper to boost::program_options::value<EnumWrapper<E>>(), this host layer's typed lookup
// (Configured<C>::getConfiguration<T>, Configured.hpp) reads any T with operator>>, so
//     Colour c = getConfiguration<EnumWrap::EnumWrapper<Colour>>("option.colour", fallback);
// plays the part of vm["option.colour"].as<EnumWrapper<Colour>>() (core/test/EnumWrapper_test.cpp:43-51).
// A token that is not in the map raises EnumWrap::validation_error, a std::logic_error as
// boost::program_options::validation_error is (EnumWrapper.hpp:104-108 translates the map's out_of_range into it).
#pragma once
#include <istream>
#include <map>
#include <stdexcept>
#include <string>

namespace EnumWrap {

class validation_error : public std::logic_error {
public:
    explicit validation_error(const std::string& token)
        : std::logic_error("the argument ('" + token + "') for the option is invalid")
    {
    }
};

template <typename E> class EnumWrapper {
public:
    typedef std::map<std::string, E> MapType;

    EnumWrapper() = default;
    EnumWrapper(E e) // lets a plain enum value serve as the default of a typed lookup
        : value(e)
    {
    }

    //! Sets and returns the wrapped value from its configuration string; std::out_of_range for an unknown key.
    E operator()(const std::string& key)
    {
        value = map().at(key);
        return value;
    }
    //! The wrapped enum as a plain one.
    operator E() const { return value; }

    //! Replaces the mapping between configuration strings and enum values.
    static void setMap(const MapType& inMap) { map() = inMap; }

    friend std::istream& operator>>(std::istream& is, EnumWrapper& e)
    {
        std::string tok;
        is >> tok;
        try {
            e(tok);
        } catch (const std::out_of_range&) {
            throw validation_error(tok);
        }
        return is;
    }

private:
    static MapType& map()
    {
        static MapType m; // one map per enum type, as the reference's static member
        return m;
    }
    E value {};
};

} // namespace EnumWrap

//! MAP_ENUM(Enum, {"x", Enum::x}, {"wye", Enum::y}) -- the helper the reference's documentation describes
//! (EnumWrapper.hpp:36-45) for filling the map in one statement.
#define MAP_ENUM(Enum, ...) EnumWrap::EnumWrapper<Enum>::setMap({ __VA_ARGS__ })
