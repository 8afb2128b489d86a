// render_framework.hpp -- the reference's backend seam (src/libs/render_framework) with a HIP
// backend behind it.
//
// Part 1 restates the abstract layer with the reference's names and behaviour: asset<> (asset.hpp:7-36),
// vertex_format / mesh / mesh_builder (meshes/mesh.hpp:23-73, mesh.cpp), parameter / parameter_builder /
// parameters_list / parameters_list_builder (paramters/parameter.hpp:23-97, parameter.cpp), texture /
// texture_builder (textures/texture.hpp), material / material_builder (materials/material.hpp:17-54),
// and utils::data (libs/utils/utils/data.hpp).
//
// Part 2 is the backend the reference writes for Vulkan (vk_mesh.cpp, vk_parameter.cpp, vk_texture.cpp,
// vk_material.cpp), written against the HIP library instead:
//   hip_mesh_builder                           vertex_format -> stride and attribute offsets
//                                              (vk_mesh.cpp:221-252), vertex / index buffers (:255-283)
//   hip_uniform_parameters_list_builder        every parameter in a slot aligned to 16 bytes
//   hip_push_constant_parameters_list_builder  (vk_parameter.cpp:12, 36-40, 226-248); a uniform list owns a
//                                              uniform buffer and flushes when dirty (:80-111), a
//                                              push-constant list is pushed at apply (:150-156)
//   hip_raw_texture_builder, hip_ktx_texture_builder   filtering / address mode -> sampler (vk_texture.cpp)
//   hip_material_builder                       stages + textures + parameter lists; apply() binds them
//
// Where the reference is unfinished the evident intent is followed and named here: the parameter
// list's byte buffer is reserved but never sized there (vk_parameter.cpp:44-55), so nothing would be
// copied; here it has its full size.  Nothing subscribes a list to its parameters there; here the
// builder does.  Shader stages are SPIR-V blobs there; the HIP library selects its three shader pairs by
// id, so a stage's code blob is the pair's name ("dummy", "test_ktx", "push_constants").
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "errors.hpp"
#include "handles.hpp"
#include "ktx2.hpp"

namespace utils {

// an owned byte range with its deleter
class data {
public:
    data() = default;
    data(uint8_t* data_ptr, size_t data_size, const std::function<void(uint8_t*)>& deleter)
        : m_data_handler(data_ptr, deleter), m_data_size(data_size) {}
    data(data&&) noexcept = default;
    data& operator=(data&&) noexcept = default;
    const uint8_t* get() const { return m_data_handler.get(); }
    uint8_t* get() { return m_data_handler.get(); }
    size_t get_size() const { return m_data_size; }
    // convenience: a copy of a byte range
    static data copy_of(const void* src, size_t size)
    {
        uint8_t* p = new uint8_t[size ? size : 1];
        if (size) std::memcpy(p, src, size);
        return data(p, size, [](uint8_t* q) { delete[] q; });
    }
private:
    std::unique_ptr<uint8_t, std::function<void(uint8_t*)>> m_data_handler{nullptr, [](uint8_t*) {}};
    size_t m_data_size{0};
};

} // namespace utils

namespace render_framework {

// ------------------------------------------------------------------ part 1: the abstract layer

template <typename AssetType, typename ImplType>
class asset {
public:
    template <typename T, typename... Args>
    static AssetType create(Args&&... args)
    {
        AssetType a{};
        a.m_impl = std::make_unique<T>(std::forward<Args>(args)...);
        return a;
    }
    asset() = default;
    asset(asset&&) noexcept = default;
    asset& operator=(asset&&) noexcept = default;
    virtual ~asset() = default;
    ImplType* get_impl() { return m_impl.get(); }
    const ImplType* get_impl() const { return m_impl.get(); }
protected:
    std::unique_ptr<ImplType> m_impl;
};

class vertex_format {
public:
    enum class attribute_type { float32, int32, int16, int8 };
    struct vertex_attribute {
        attribute_type type;
        size_t elements_count;
    };
    void add_attribute(vertex_attribute a) { m_attributes.push_back(a); }
    const std::vector<vertex_attribute>& get_attributes() const { return m_attributes; }
private:
    std::vector<vertex_attribute> m_attributes;
};

namespace detail {
class mesh_impl {
public:
    virtual ~mesh_impl() = default;
    virtual const vertex_format& get_format() const = 0;
};
} // namespace detail

class mesh : public asset<mesh, detail::mesh_impl> {
public:
    const vertex_format& get_format() const { return m_impl->get_format(); }
};

class mesh_builder {
public:
    enum class index_type { int32, int16, int8 };
    virtual ~mesh_builder() = default;
    mesh_builder& set_format(const vertex_format& f) { m_vertex_format.emplace(f); return *this; }
    mesh_builder& set_index_format(index_type t) { m_index_format = t; return *this; }
    mesh_builder& set_vertex_data(utils::data d) { m_vertex_data = std::move(d); return *this; }
    mesh_builder& set_index_data(utils::data d) { m_index_data = std::move(d); return *this; }
    virtual ERROR_TYPE create(mesh&) = 0;
protected:
    virtual void clear()
    {
        m_vertex_data = utils::data{};
        m_index_data = utils::data{};
        m_vertex_format.reset();
    }
    utils::data m_vertex_data{};
    utils::data m_index_data{};
    std::optional<vertex_format> m_vertex_format{};
    index_type m_index_format{};
};

class parameter;

class parameter_onupdate_listenner {
public:
    virtual ~parameter_onupdate_listenner() = default;
    virtual void on_parameter_updated(const parameter* parameter, const uint8_t* value) = 0;
};

class parameter {
    friend class parameter_builder;
public:
    enum class value_type { empty, int32, float32, vec4, mat4 };
    parameter() = default;
    void subscribe_onupdate_listenner(parameter_onupdate_listenner* l) { m_listenners.push_back(l); }
    void unsubscribe_onupdate_listenner(const parameter_onupdate_listenner* l)
    {
        m_listenners.erase(std::remove(m_listenners.begin(), m_listenners.end(), l), m_listenners.end());
    }
    void load_value(const uint8_t* value) { for (auto* l : m_listenners) l->on_parameter_updated(this, value); }
    value_type get_value_type() const { return m_value_type; }
    size_t get_elements_count() const { return m_elements_count; }
private:
    parameter(value_type t, size_t n) : m_value_type(t), m_elements_count(n) {}
    std::vector<parameter_onupdate_listenner*> m_listenners;
    value_type m_value_type = value_type::empty;
    size_t m_elements_count{0};
};

class parameter_builder {
public:
    virtual ~parameter_builder() = default;
    parameter_builder& set_value_type(parameter::value_type t) { m_value_type = t; return *this; }
    parameter_builder& set_elements_count(size_t n) { m_elements_count = n; return *this; }
    ERROR_TYPE create(parameter& res)
    {
        if (m_elements_count == 0) RAISE_ERROR_WARN(-1, "bad elements count.");
        if (m_value_type == parameter::value_type::empty) RAISE_ERROR_WARN(-1, "bad parameter value.");
        res = parameter(m_value_type, m_elements_count);
        RAISE_ERROR_OK();
    }
protected:
    parameter::value_type m_value_type = parameter::value_type::vec4;
    size_t m_elements_count{1};
};

class parameters_list_impl : public parameter_onupdate_listenner {
public:
    ~parameters_list_impl() override = default;
};

class parameters_list : public asset<parameters_list, parameters_list_impl> {};

class parameters_list_builder {
public:
    virtual ~parameters_list_builder() = default;
    parameters_list_builder& add_parameter(parameter& p) { m_parameters_list.push_back(&p); return *this; }
    virtual ERROR_TYPE create(parameters_list&) = 0;
protected:
    std::vector<parameter*> m_parameters_list;
};

class texture_impl {
public:
    virtual ~texture_impl() = default;
};

class texture : public asset<texture, texture_impl> {};

class texture_builder {
public:
    enum class texture_filtering { point, linear, bilinear, anizotropic };
    enum class texture_address_mode { repeat, mirror_repeat, clamp_to_egde, clamp_to_border, mirror_clamp_to_edge };
    virtual ~texture_builder() = default;
    texture_builder& set_filtering(texture_filtering f) { m_filtering = f; return *this; }
    texture_builder& set_address_mode(texture_address_mode m) { m_address_mode = m; return *this; }
    virtual ERROR_TYPE create(texture& result) = 0;
protected:
    texture_filtering m_filtering = texture_filtering::point;
    texture_address_mode m_address_mode = texture_address_mode::clamp_to_egde;
};

namespace detail {
class material_impl {
public:
    virtual ~material_impl() = default;
    virtual void apply() = 0;
    virtual void set_texture(size_t slot, const texture& texture) = 0;
};
} // namespace detail

class material : public asset<material, detail::material_impl> {
public:
    void apply() { m_impl->apply(); }
    void set_texture(size_t slot, const texture& t) { m_impl->set_texture(slot, t); }
};

class material_builder {
public:
    enum class shading_stage_type { vertex, fragment };
    virtual ~material_builder() = default;
    material_builder& add_stage(shading_stage_type s, utils::data code) { m_stages[s] = std::move(code); return *this; }
    material_builder& add_texture(const texture& t) { m_textures.push_back(&t); return *this; }
    material_builder& add_parameters_list(const parameters_list& p) { m_parameters.push_back(&p); return *this; }
    virtual ERROR_TYPE create(material&) = 0;
protected:
    virtual void clear() { m_stages.clear(); m_textures.clear(); m_parameters.clear(); }
    struct stage_hash { size_t operator()(shading_stage_type s) const { return (size_t)s; } };
    std::unordered_map<shading_stage_type, utils::data, stage_hash> m_stages;
    std::vector<const texture*> m_textures;
    std::vector<const parameters_list*> m_parameters;
};

// ------------------------------------------------------------------ part 2: the HIP backend

namespace detail {

// what the draw path needs of a vertex format: where position, normal and uv sit, and the stride
struct hip_vertex_layout {
    uint32_t stride{0};
    rnd_vertex_attribute position{0, 0}, normal{0, 0}, uv{0, 0};
};

class hip_mesh_impl : public mesh_impl {
public:
    hip_mesh_impl(vertex_format f, hip_vertex_layout layout, rnd_host::buffer_handler vb, rnd_host::buffer_handler ib, uint32_t index_count)
        : m_format(std::move(f)), m_layout(layout), m_vertex_buffer(std::move(vb)), m_index_buffer(std::move(ib)), m_index_count(index_count) {}
    const vertex_format& get_format() const override { return m_format; }
    const hip_vertex_layout& get_layout() const { return m_layout; }
    rnd_buffer get_vertex_buffer() const { return m_vertex_buffer; }
    rnd_buffer get_index_buffer() const { return m_index_buffer; }
    uint32_t get_index_count() const { return m_index_count; }
private:
    vertex_format m_format;
    hip_vertex_layout m_layout;
    rnd_host::buffer_handler m_vertex_buffer, m_index_buffer;
    uint32_t m_index_count;
};

} // namespace detail

class hip_mesh_builder : public mesh_builder {
public:
    explicit hip_mesh_builder(rnd_device dev) : m_device(dev) {}

    ERROR_TYPE create(mesh& result) override
    {
        detail::hip_vertex_layout layout;
        PASS_ERROR(create_vertex_inputs(layout));
        if (m_vertex_data.get() == nullptr) RAISE_ERROR_WARN(-1, "vertex data wasn't settled.");
        if (m_index_data.get() == nullptr) RAISE_ERROR_WARN(-1, "index data wasn't settled: the draw path is indexed.");
        if (m_vertex_data.get_size() % layout.stride != 0) RAISE_ERROR_WARN(-1, "vertex data is not a whole number of vertices.");
        // the library draws with 32-bit indices (the reference's obj path, obj_loader.cpp:198-200): narrower ones are widened
        std::vector<uint32_t> indices;
        const uint8_t* src = m_index_data.get();
        switch (m_index_format) {
        case index_type::int32:
            indices.resize(m_index_data.get_size() / 4);
            std::memcpy(indices.data(), src, indices.size() * 4);
            break;
        case index_type::int16:
            indices.resize(m_index_data.get_size() / 2);
            for (size_t i = 0; i < indices.size(); ++i) { uint16_t v; std::memcpy(&v, src + 2 * i, 2); indices[i] = v; }
            break;
        case index_type::int8:
            indices.resize(m_index_data.get_size());
            for (size_t i = 0; i < indices.size(); ++i) indices[i] = src[i];
            break;
        }
        if (indices.empty() || indices.size() % 3 != 0) RAISE_ERROR_WARN(-1, "index data is not a whole number of triangles.");
        rnd_host::buffer_handler vb, ib;
        RND_CHECK(rnd_buffer_create(m_device, m_vertex_data.get_size(), RND_BUFFER_USAGE_VERTEX, m_vertex_data.get(), vb.put()));
        RND_CHECK(rnd_buffer_create(m_device, indices.size() * 4, RND_BUFFER_USAGE_INDEX, indices.data(), ib.put()));
        result = mesh::create<detail::hip_mesh_impl>(*m_vertex_format, layout, std::move(vb), std::move(ib), (uint32_t)indices.size());
        clear();
        RAISE_ERROR_OK();
    }

    // attribute i sits at the sum of the sizes before it; the stride is the sum of all (vk_mesh.cpp:221-252).
    // The shaders read a float3 position first, then optionally a float3 normal and a float2 uv.
    ERROR_TYPE create_vertex_inputs(detail::hip_vertex_layout& out) const
    {
        if (!m_vertex_format.has_value()) RAISE_ERROR_WARN(-1, "vertex format wasn't settled.");
        uint32_t offset = 0;
        int float3_seen = 0;
        for (const auto& a : m_vertex_format->get_attributes()) {
            if (a.elements_count < 1 || a.elements_count > 4) RAISE_ERROR_WARN(-1, "bad elements count");
            uint32_t size = 0;
            switch (a.type) {
            case vertex_format::attribute_type::float32: size = 4u * (uint32_t)a.elements_count; break;
            case vertex_format::attribute_type::int32: size = 4u * (uint32_t)a.elements_count; break;
            case vertex_format::attribute_type::int16: size = 2u * (uint32_t)a.elements_count; break;
            case vertex_format::attribute_type::int8: size = 1u * (uint32_t)a.elements_count; break;
            }
            if (a.type == vertex_format::attribute_type::float32 && a.elements_count == 3) {
                if (float3_seen == 0) out.position = {RND_FORMAT_R32G32B32_SFLOAT, offset};
                else if (float3_seen == 1) out.normal = {RND_FORMAT_R32G32B32_SFLOAT, offset};
                ++float3_seen;
            } else if (a.type == vertex_format::attribute_type::float32 && a.elements_count == 2 && out.uv.format == 0) {
                out.uv = {RND_FORMAT_R32G32_SFLOAT, offset};
            }
            offset += size;
        }
        if (out.position.format == 0) RAISE_ERROR_WARN(-1, "vertex format has no float3 position.");
        if (offset % 4 != 0) RAISE_ERROR_WARN(-1, "vertex stride must be a multiple of 4 bytes.");
        out.stride = offset;
        RAISE_ERROR_OK();
    }

private:
    rnd_device m_device;
};

namespace detail {

class hip_uniform_parameters_list_impl;
class hip_push_constant_parameters_list_impl;

class hip_parameters_list_visitor {
public:
    virtual ~hip_parameters_list_visitor() = default;
    virtual void accept(const hip_uniform_parameters_list_impl*) const = 0;
    virtual void accept(const hip_push_constant_parameters_list_impl*) const = 0;
};

class hip_parameters_list_impl : public parameters_list_impl {
public:
    struct parameter_data {
        size_t data_offset;
        size_t data_size;
    };
    hip_parameters_list_impl(std::vector<parameter_data> parameters, std::unordered_map<const parameter*, size_t> map, size_t buffer_size)
        : m_parameters_data(std::move(parameters)), m_parameters_map(std::move(map)), m_data_buffer(buffer_size, 0) {}
    void on_parameter_updated(const parameter* p, const uint8_t* value) override
    {
        const auto it = m_parameters_map.find(p);
        if (it == m_parameters_map.end()) return;
        const parameter_data& view = m_parameters_data[it->second];
        std::memcpy(m_data_buffer.data() + view.data_offset, value, view.data_size);
        m_dirty = true;
    }
    virtual void visit(const hip_parameters_list_visitor* visitor) const = 0;
    size_t get_buffer_offset() const { return 0; }
    size_t get_buffer_size() const { return m_data_buffer.size(); }
    const uint8_t* get_data() const { return m_data_buffer.data(); }
    const std::vector<parameter_data>& get_parameters_data() const { return m_parameters_data; }
protected:
    std::vector<parameter_data> m_parameters_data;
    std::unordered_map<const parameter*, size_t> m_parameters_map;
    std::vector<uint8_t> m_data_buffer;
    mutable bool m_dirty{true};
};

class hip_uniform_parameters_list_impl : public hip_parameters_list_impl {
public:
    hip_uniform_parameters_list_impl(std::vector<parameter_data> parameters, std::unordered_map<const parameter*, size_t> map, size_t buffer_size,
                                     rnd_host::buffer_handler uniform_buffer)
        : hip_parameters_list_impl(std::move(parameters), std::move(map), buffer_size), m_uniform_buffer(std::move(uniform_buffer)) {}
    // staging copy + vkCmdCopyBuffer in the reference (vk_parameter.cpp:80-111); one buffer write here
    int flush() const
    {
        if (!m_dirty) return RND_SUCCESS;
        const int rc = rnd_buffer_write(m_uniform_buffer, 0, m_data_buffer.data(), m_data_buffer.size());
        if (rc == RND_SUCCESS) m_dirty = false;
        return rc;
    }
    void visit(const hip_parameters_list_visitor* visitor) const override { visitor->accept(this); }
    rnd_buffer get_buffer() const { return m_uniform_buffer; }
private:
    rnd_host::buffer_handler m_uniform_buffer;
};

class hip_push_constant_parameters_list_impl : public hip_parameters_list_impl {
public:
    using hip_parameters_list_impl::hip_parameters_list_impl;
    // vkCmdPushConstants(cmd, layout, stages, 0, size, data) (vk_parameter.cpp:150-156)
    int flush(rnd_cmd command_buffer) const
    {
        return rnd_cmd_push_constants(command_buffer, 0, (uint32_t)m_data_buffer.size(), m_data_buffer.data());
    }
    void visit(const hip_parameters_list_visitor* visitor) const override { visitor->accept(this); }
};

} // namespace detail

class hip_parameters_list_builder : public parameters_list_builder {
protected:
    struct parameters_list_args {
        std::vector<detail::hip_parameters_list_impl::parameter_data> params_data;
        std::unordered_map<const parameter*, size_t> parameters_map;
        size_t buffer_size{0};
    };
    static constexpr size_t buffer_alignment = sizeof(float) * 4;

    // vk_parameter.cpp:226-248: one slot per parameter, slots aligned to 16 bytes; the element
    // count does not enter the size there either
    ERROR_TYPE get_paramters_list_args(parameters_list_args& args) const
    {
        for (const parameter* p : m_parameters_list) {
            size_t value_size = 0;
            switch (p->get_value_type()) {
            case parameter::value_type::int32: value_size = sizeof(int32_t); break;
            case parameter::value_type::float32: value_size = sizeof(float); break;
            case parameter::value_type::vec4: value_size = sizeof(float) * 4; break;
            case parameter::value_type::mat4: value_size = sizeof(float) * 16; break;
            default: RAISE_ERROR_WARN(-1, "bad value type.");
            }
            args.parameters_map[p] = args.params_data.size();
            args.params_data.push_back({args.buffer_size, value_size});
            args.buffer_size += (value_size + buffer_alignment - 1) / buffer_alignment * buffer_alignment;
        }
        if (args.buffer_size == 0) RAISE_ERROR_WARN(-1, "parameters list is empty.");
        RAISE_ERROR_OK();
    }
    void subscribe(parameters_list& list)
    {
        for (parameter* p : m_parameters_list) p->subscribe_onupdate_listenner(list.get_impl());
        m_parameters_list.clear();
    }
};

class hip_uniform_parameters_list_builder : public hip_parameters_list_builder {
public:
    explicit hip_uniform_parameters_list_builder(rnd_device dev) : m_device(dev) {}
    ERROR_TYPE create(parameters_list& result) override
    {
        parameters_list_args args;
        PASS_ERROR(get_paramters_list_args(args));
        rnd_host::buffer_handler uniform_buffer;
        RND_CHECK(rnd_buffer_create(m_device, args.buffer_size, RND_BUFFER_USAGE_UNIFORM, nullptr, uniform_buffer.put()));
        result = parameters_list::create<detail::hip_uniform_parameters_list_impl>(std::move(args.params_data), std::move(args.parameters_map),
                                                                                    args.buffer_size, std::move(uniform_buffer));
        subscribe(result);
        RAISE_ERROR_OK();
    }
private:
    rnd_device m_device;
};

class hip_push_constant_parameters_list_builder : public hip_parameters_list_builder {
public:
    ERROR_TYPE create(parameters_list& result) override
    {
        parameters_list_args args;
        PASS_ERROR(get_paramters_list_args(args));
        if (args.buffer_size > 128) RAISE_ERROR_WARN(-1, "push constants are limited to 128 bytes.");
        result = parameters_list::create<detail::hip_push_constant_parameters_list_impl>(std::move(args.params_data), std::move(args.parameters_map),
                                                                                          args.buffer_size);
        subscribe(result);
        RAISE_ERROR_OK();
    }
};

namespace detail {
class hip_texture_impl : public texture_impl {
public:
    explicit hip_texture_impl(rnd_host::texture_handler t) : m_texture(std::move(t)) {}
    rnd_texture get_texture() const { return m_texture; }
private:
    rnd_host::texture_handler m_texture;
};
} // namespace detail

class hip_texture_builder : public texture_builder {
public:
    explicit hip_texture_builder(rnd_device dev) : m_device(dev) {}
    ERROR_TYPE create(texture& result) override
    {
        rnd_sampler_info sampler{};
        sampler.filtering = m_filtering == texture_filtering::point ? RND_FILTER_NEAREST : RND_FILTER_LINEAR;
        if (m_filtering == texture_filtering::anizotropic) RAISE_ERROR_WARN(-1, "anisotropic filtering is not supported.");
        if (m_address_mode == texture_address_mode::repeat) sampler.tiled = 1;
        else if (m_address_mode == texture_address_mode::clamp_to_egde) sampler.tiled = 0;
        else RAISE_ERROR_WARN(-1, "address mode must be repeat or clamp_to_egde.");
        rnd_host::texture_handler t;
        PASS_ERROR(create_hip_texture(sampler, t));
        result = texture::create<detail::hip_texture_impl>(std::move(t));
        RAISE_ERROR_OK();
    }
protected:
    virtual ERROR_TYPE create_hip_texture(const rnd_sampler_info& sampler, rnd_host::texture_handler& out) = 0;
    rnd_device m_device;
};

// texels already decoded (the place of the reference's stb builder: PNG / JPG decoding is a third-party library)
class hip_raw_texture_builder : public hip_texture_builder {
public:
    using hip_texture_builder::hip_texture_builder;
    hip_raw_texture_builder& set_texels(uint32_t width, uint32_t height, int32_t format, bool gen_mips, utils::data texels)
    {
        m_width = width; m_height = height; m_format = format; m_gen_mips = gen_mips; m_texels = std::move(texels);
        return *this;
    }
protected:
    ERROR_TYPE create_hip_texture(const rnd_sampler_info& sampler, rnd_host::texture_handler& out) override
    {
        if (m_texels.get() == nullptr) RAISE_ERROR_WARN(-1, "texture data wasn't settled.");
        RND_CHECK(rnd_texture_create_2d(m_device, m_width, m_height, m_format, m_gen_mips ? 1 : 0, m_texels.get(), &sampler, out.put()));
        RAISE_ERROR_OK();
    }
private:
    uint32_t m_width{0}, m_height{0};
    int32_t m_format{RND_FORMAT_R8G8B8A8_SRGB};
    bool m_gen_mips{true};
    utils::data m_texels{};
};

class hip_ktx_texture_builder : public hip_texture_builder {
public:
    using hip_texture_builder::hip_texture_builder;
    hip_ktx_texture_builder& set_file_data(utils::data file) { m_file = std::move(file); return *this; }
protected:
    ERROR_TYPE create_hip_texture(const rnd_sampler_info& sampler, rnd_host::texture_handler& out) override
    {
        if (m_file.get() == nullptr) RAISE_ERROR_WARN(-1, "texture data wasn't settled.");
        PASS_ERROR(rnd_host::create_ktx_texture(m_device, m_file.get(), m_file.get_size(), sampler, out));
        RAISE_ERROR_OK();
    }
private:
    utils::data m_file{};
};

namespace detail {

class hip_material_impl : public material_impl, public hip_parameters_list_visitor {
public:
    hip_material_impl(int32_t shader, std::vector<const texture*> textures, std::vector<const parameters_list*> lists)
        : m_shader(shader), m_parameters_lists(std::move(lists))
    {
        for (size_t i = 0; i < textures.size(); ++i) m_textures[i] = textures[i];
    }
    // descriptor writes, uniform flushes and push constants (vk_material.cpp apply): on the command
    // buffer given to begin_apply(); apply() without one only flushes the uniform lists
    void apply() override { apply(nullptr); }
    int apply(rnd_cmd command_buffer)
    {
        m_command_buffer = command_buffer;
        m_result = RND_SUCCESS;
        for (const parameters_list* l : m_parameters_lists)
            static_cast<const hip_parameters_list_impl*>(l->get_impl())->visit(this);
        if (command_buffer && m_result == RND_SUCCESS) {
            const auto it = m_textures.find(0);
            const rnd_texture t = it == m_textures.end() || !it->second ? nullptr : static_cast<const hip_texture_impl*>(it->second->get_impl())->get_texture();
            m_result = rnd_cmd_bind_texture(command_buffer, 0, 1, t);
        }
        m_command_buffer = nullptr;
        return m_result;
    }
    void set_texture(size_t slot, const texture& t) override { m_textures[slot] = &t; }
    void accept(const hip_uniform_parameters_list_impl* impl) const override
    {
        if (m_result != RND_SUCCESS) return;
        m_result = impl->flush();
        if (m_command_buffer && m_result == RND_SUCCESS)
            m_result = rnd_cmd_bind_uniform(m_command_buffer, 0, 0, impl->get_buffer(), impl->get_buffer_offset(), impl->get_buffer_size());
    }
    void accept(const hip_push_constant_parameters_list_impl* impl) const override
    {
        if (m_result != RND_SUCCESS || !m_command_buffer) return;
        m_result = impl->flush(m_command_buffer);
    }
    int32_t get_shader() const { return m_shader; }
private:
    int32_t m_shader;
    std::vector<const parameters_list*> m_parameters_lists;
    std::unordered_map<size_t, const texture*> m_textures;
    rnd_cmd m_command_buffer{nullptr};
    mutable int m_result{RND_SUCCESS};
};

} // namespace detail

class hip_material_builder : public material_builder {
public:
    ERROR_TYPE create(material& result) override
    {
        const auto vs = m_stages.find(shading_stage_type::vertex), fs = m_stages.find(shading_stage_type::fragment);
        if (vs == m_stages.end() || fs == m_stages.end()) RAISE_ERROR_WARN(-1, "vertex and fragment stages are required.");
        const std::string vname(reinterpret_cast<const char*>(vs->second.get()), vs->second.get_size());
        const std::string fname(reinterpret_cast<const char*>(fs->second.get()), fs->second.get_size());
        if (vname != fname) RAISE_ERROR_WARN(-1, "vertex and fragment stages must name the same shader pair.");
        int32_t shader;
        if (vname == "dummy") shader = RND_SHADER_DUMMY;
        else if (vname == "test_ktx") shader = RND_SHADER_TEST_KTX;
        else if (vname == "push_constants") shader = RND_SHADER_PUSH_CONSTANTS;
        else RAISE_ERROR_WARN(-1, "unknown shader pair '" + vname + "'.");
        size_t uniform_lists = 0;
        for (const parameters_list* l : m_parameters) {
            if (!l || !l->get_impl()) RAISE_ERROR_WARN(-1, "parameters list wasn't created.");
            if (dynamic_cast<const detail::hip_uniform_parameters_list_impl*>(l->get_impl())) ++uniform_lists;
        }
        if (uniform_lists != 1) RAISE_ERROR_WARN(-1, "the shaders read exactly one uniform block (set 0, binding 0).");
        result = material::create<detail::hip_material_impl>(shader, m_textures, m_parameters);
        clear();
        RAISE_ERROR_OK();
    }
};

// The pipeline a mesh and a material draw with: the samples' fixed-function state
// (test_ktx_app.cpp:435-499) around the mesh's vertex layout and the material's shader pair.
inline ERROR_TYPE create_pipeline(rnd_device dev, const mesh& m, const material& mat, uint32_t samples, rnd_host::pipeline_handler& out)
{
    const auto* mesh_impl = static_cast<const detail::hip_mesh_impl*>(m.get_impl());
    const auto* mat_impl = static_cast<const detail::hip_material_impl*>(mat.get_impl());
    if (!mesh_impl || !mat_impl) RAISE_ERROR_WARN(-1, "mesh or material wasn't created.");
    rnd_pipeline_info info{};
    info.shader = mat_impl->get_shader();
    info.position = mesh_impl->get_layout().position;
    info.normal = mesh_impl->get_layout().normal;
    info.uv = mesh_impl->get_layout().uv;
    info.stride = mesh_impl->get_layout().stride;
    info.topology = RND_TOPOLOGY_TRIANGLE_LIST;
    info.cull_mode = RND_CULL_BACK;
    info.front_face = RND_FRONT_FACE_COUNTER_CLOCKWISE;
    info.depth_test = 1; info.depth_write = 1; info.depth_compare = RND_COMPARE_LESS_OR_EQUAL;
    info.samples = samples;
    info.color_write_mask = 0xF;
    RND_CHECK(rnd_pipeline_create(dev, &info, out.put()));
    RAISE_ERROR_OK();
}

// bind + draw of a mesh inside a pass
inline ERROR_TYPE draw_mesh(rnd_cmd cmd, const mesh& m)
{
    const auto* impl = static_cast<const detail::hip_mesh_impl*>(m.get_impl());
    if (!impl) RAISE_ERROR_WARN(-1, "mesh wasn't created.");
    RND_CHECK(rnd_cmd_bind_vertex_buffer(cmd, impl->get_vertex_buffer(), 0));
    RND_CHECK(rnd_cmd_bind_index_buffer(cmd, impl->get_index_buffer(), 0, RND_INDEX_TYPE_UINT32));
    RND_CHECK(rnd_cmd_draw_indexed(cmd, impl->get_index_count(), 1, 0, 0, 0));
    RAISE_ERROR_OK();
}

} // namespace render_framework
